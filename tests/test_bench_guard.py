"""CPU: bench.py's _LineGuard -- whatever happens to the inference leg (an exception on rank 0, another rank's death = SIGTERM from the
launcher, a hang), rank 0 still prints the measured train line, with `inference: {error: ...}`, and leaves non-zero.
Each case runs in a child process: the guard ends its process with os._exit."""
import json
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(body):
    code = textwrap.dedent('''
        import os, signal, sys, time
        sys.path.insert(0, %r)
        import bench
        out = dict(metric='voxels/sec', value=123.0, ms_per_step=27.5)
    ''' % ROOT) + textwrap.dedent(body)
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=120)
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    return r.returncode, [json.loads(l) for l in lines], r.stderr


def test_exception_on_rank0_keeps_the_train_line():
    rc, lines, err = _run('''
        g = bench._LineGuard(out, 0, 1, 60.0)
        try:
            raise RuntimeError('boom')
        except Exception as e:
            g.fail('inference leg raised on rank 0: %r' % (e,))
    ''')
    assert rc == 3 and len(lines) == 1
    assert lines[0]['value'] == 123.0 and 'boom' in lines[0]['inference']['error']


def test_sigterm_while_blocked_prints_the_line():
    rc, lines, err = _run('''
        import threading
        g = bench._LineGuard(out, 0, 2, 60.0)
        threading.Timer(0.3, lambda: os.kill(os.getpid(), signal.SIGTERM)).start()
        threading.Event().wait(30)   # the main thread is parked (as inside a collective); the helper thread must do the printing
        time.sleep(30)
    ''')
    assert rc == 3 and len(lines) == 1, (rc, lines, err)
    assert 'another rank failed' in lines[0]['inference']['error'] and lines[0]['ms_per_step'] == 27.5


def test_hang_is_cut_by_the_watchdog():
    rc, lines, err = _run('''
        g = bench._LineGuard(out, 0, 2, 0.5)
        time.sleep(30)
    ''')
    assert rc == 3 and len(lines) == 1
    assert 'watchdog' in lines[0]['inference']['error']


def test_failing_rank_above_zero_prints_nothing_and_leaves_nonzero():
    rc, lines, err = _run('''
        g = bench._LineGuard(out, 1, 2, 60.0)
        g.fail('inference leg raised on rank 1: x')
    ''')
    assert rc == 3 and lines == [] and 'rank 1' in err


def test_disarm_restores_normal_exit():
    rc, lines, err = _run('''
        g = bench._LineGuard(out, 0, 2, 60.0)
        g.disarm()
        time.sleep(0.2)
        print('{"ok": true}')
    ''')
    assert rc == 0 and lines == [{'ok': True}]
