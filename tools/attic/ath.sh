for v in 3 1 3 1 0; do
echo "athena NC_P2D=$v $(NC_P2D=$v timeout 300 python bench.py --workload train --model athena --data structured --steps 8 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.readline()); print('%.2f ms' % j['ms_per_step'], j['config']['first_step_losses']['G_A'], j['config']['first_step_losses']['D_B_xz'])")"
done
timeout 900 python -m pytest tests/test_gpu_nets.py tests/test_gpu_fullsize.py tests/test_gpu_structured.py tests/test_gpu_p2d.py -x -q -m gpu -k "athena or patchgan or discriminators or p2d" 2>&1 | tail -4
