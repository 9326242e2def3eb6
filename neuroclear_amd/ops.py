"""torch.autograd.Function wrappers over the C ABI (include/nc_hip.h).  PyTorch provides device memory, streams and
the autograd tape only -- every forward/backward below is one or more HIP kernels from libnc_hip.so.

Each op mirrors the torch.nn call the reference makes on its hot path (models/networks.py, apollo_model.py); see the
header for the file:line of every call site."""
import ctypes
import os

import torch

from . import _lib
from ._lib import F, I, L_, P, Z, check, lib

_ws_cache = {}

# Live profiler (bench.py).  Convolution launches are bracketed INSIDE the library (nc_prof_begin / nc_prof_end: HIP
# events on the launch stream around every convolution entry point, also those issued by the whole-network calls);
# `prof` additionally collects (tag, algorithmic FLOP, start, end) around the whole-network C calls themselves.
prof = None


def prof_start(min_flop=0.0):
    global prof
    prof = []
    lib().nc_prof_begin(ctypes.c_double(min_flop))


_PATH = {0: 'direct', 1: 'mfma', 2: 'gemm', 3: 'flat', 4: 'taps', 5: 'k1', 6: 'to1', 7: 'img', 8: 'pg1', 9: 'split', 10: 'split2d', 11: 'split'}


def prof_stop():
    """-> ({tag: [launches, ms, flop, algorithmic bytes]} of the convolution launches, [(tag, flop, ms)] of the whole-network calls).  The
    caller has synchronised the device."""
    global prof
    L = lib()
    cap = 1 << 16
    cls, flop, ms, ab = (ctypes.c_int * cap)(), (ctypes.c_double * cap)(), (ctypes.c_float * cap)(), (ctypes.c_double * cap)()
    n = min(L.nc_prof_end2(I(cap), cls, flop, ms, ab), cap)
    stats = {}
    for i in range(n):
        c = cls[i]
        op, path, k, lp = ('fwd', 'dgrad', 'wgrad')[c & 15], (c >> 4) & 15, (c >> 8) & 255, (c >> 16) & 1
        tag = '%s_lp_k%d' % (op, k) if lp else '%s_%s_k%d' % (op, _PATH.get(path, '?'), k)
        s = stats.setdefault(tag, [0, 0.0, 0.0, 0.0])
        s[0] += 1
        s[1] += ms[i]
        s[2] += flop[i]
        s[3] += ab[i]  # algorithmic bytes (operands once + result + weights)
    whole = [(tag, fl, e0.elapsed_time(e1)) for tag, fl, e0, e1 in (prof or [])]
    prof = None
    return stats, whole


def _prof_begin(flop=None):
    if prof is None or flop is not None:  # per-convolution brackets live in the library
        return None
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    return e


def _prof_end(e0, tag, flop):
    if e0 is not None:
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        prof.append((tag, flop, e0, e1))


def _conv_tag(op, C, K, k3, stride, pad, out_vox):
    mf = lib().nc_conv_wgrad_path if op == 'wgrad' else lib().nc_conv_fwd_path
    if op == 'dgrad':
        path = lib().nc_conv_fwd_path(I(K), I(C), I(k3[0]), I(k3[1]), I(k3[2]), I(stride), I(pad))
    else:
        path = mf(I(C), I(K), I(k3[0]), I(k3[1]), I(k3[2]), I(stride), I(pad))
    tag = '%s_%s_k%d' % (op, {1: 'mfma', 2: 'gemm', 3: 'flat', 4: 'taps', 9: 'split'}.get(path, 'direct'), k3[1])
    return tag, 2.0 * C * K * k3[0] * k3[1] * k3[2] * out_vox


def _stream():
    return P(torch.cuda.current_stream().cuda_stream)


def _chk(*ts):
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise _lib.NcError('neuroclear_amd ops need CUDA(HIP) tensors; there is no CPU fallback')
        if not t.is_contiguous():
            raise _lib.NcError('tensor must be contiguous')


def _f32(*ts):
    for t in ts:
        if t is not None and t.dtype != torch.float32:
            raise _lib.NcError('expected float32, got %s' % t.dtype)


def _ptr(t):
    return P(t.data_ptr()) if t is not None else P(0)


def workspace(nbytes, device, tag='ws'):
    """Grow-only scratch buffer per (device, tag, current stream): reuse is stream-ordered, and independent networks
    that run concurrently on different streams (the discriminators of the Apollo step) never share scratch."""
    key = (str(device), tag, torch.cuda.current_stream(device).cuda_stream)
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        if not _ws_cache:
            # first allocation of the process: also the moment the library takes its pinned counter block of the range guard (a hipHostMalloc
            # that must not happen on the launch path, where it could land inside a stream capture: csrc/h2.hip)
            lib().nc_set_h2_guard(lib().nc_get_h2_guard())
        buf = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)
        _ws_cache[key] = buf
    return buf


def _dims5(shape):
    if len(shape) == 5:
        return tuple(shape)
    if len(shape) == 4:
        n, c, h, w = shape
        return n, c, 1, h, w
    raise _lib.NcError('expected NCDHW or NCHW tensor')


def _kdims(wshape):
    if len(wshape) == 5:
        return wshape[2], wshape[3], wshape[4]
    return 1, wshape[2], wshape[3]


def _conv_out_shape(xs, ws, stride, pad):
    nsp = len(xs) - 2
    return (xs[0], ws[0]) + tuple((xs[2 + i] + 2 * pad - ws[2 + i]) // stride + 1 for i in range(nsp))


def _conv_ws(dims, K, k3, stride, pad, device, tag='ws'):
    N, C, D, H, W = dims
    nb = lib().nc_conv_ws_bytes(I(N), I(C), I(D), I(H), I(W), I(K), I(k3[0]), I(k3[1]), I(k3[2]), I(stride), I(pad))
    return workspace(nb, device, tag)


# Backward runs the weight gradient of a layer on a side stream, concurrently with the data gradient of the same layer
# (they are independent given dy): the two kernels' workgroups interleave on the 256 CUs, which fills the idle tail
# each of them leaves when its tile count is not a multiple of the CU count (e.g. 1296 tiles = 5.06 rounds at 108^3).
# NC_WGRAD_STREAM=1: weight gradients of the big layers on a side stream behind the data gradient.  Off by default:
# once both kernels fill the chip by themselves (round-1 end state) running them side by side is 0.3 % slower.
overlap_wgrad = os.environ.get('NC_WGRAD_STREAM', '0') != '0'
_side_streams = {}


def _side_stream(device):
    st = _side_streams.get(device)
    if st is None:
        st = _side_streams[device] = torch.cuda.Stream(device=device)
    return st


# Arithmetic of the convolutions (BASELINE.json configs[3]): 'fp32' (default, the reference's arithmetic) or 'bf16' /
# 'fp16' = operands rounded to 16 bits, fp32 accumulation, on the layers nc_conv_lp_supported covers (3^3 / 5^3, stride
# 1, >= 16 input and >= 64 output channels); every other layer and everything between the convolutions stays fp32.
_DT = {'fp32': 0, 'fp16': 1, 'bf16': 2}
conv_precision = 'fp32'


def set_conv_split(on):
    """fp32 3^3 / 5^3 convolutions as six bf16 MFMA products of an exact three-term operand split (csrc/conv_split.hip; the
    library default) or, with on=False, on the fp32 MFMA kernels.  Returns the previous setting."""
    prev = bool(lib().nc_get_conv_split())
    lib().nc_set_conv_split(I(1 if on else 0))
    return prev


def set_conv_precision(name):
    global conv_precision
    if name not in _DT:
        raise _lib.NcError('conv precision must be one of %s, got %r' % (sorted(_DT), name))
    conv_precision = name


_lp_cache = {}


def _lp(what, dims, K, k3, stride, pad):
    """dtype code of the 16-bit kernel for this call, or 0 when the fp32 kernels serve it."""
    if conv_precision == 'fp32':
        return 0
    key = (what, dims, K, k3, stride, pad)
    ok = _lp_cache.get(key)
    if ok is None:
        N, C, D, H, W = dims
        ok = _lp_cache[key] = bool(lib().nc_conv_lp_supported(I(what), I(N), I(C), I(D), I(H), I(W), I(K), I(k3[0]),
                                                              I(k3[1]), I(k3[2]), I(stride), I(pad)))
    if not ok:
        return 0
    # 'fp16': forward operands in fp16 (11-bit significand); backward operands (dy, and w / x next to it) in bf16 --
    # gradients of a mean loss over 1e6..1e7 voxels sit below fp16's normal range (6e-5) and there is no loss scaling
    # in the reference's step to lean on; bf16 has fp32's exponent range (measured: tools/lp_err.py).
    if conv_precision == 'fp16' and what != 0:
        return _DT['bf16']
    return _DT[conv_precision]


def to_c8(x, dt):
    """fp32 NCDHW -> the 16-bit C8 operand layout of the *_lp kernels (include/nc_hip.h); returns a byte tensor."""
    _chk(x)
    _f32(x)
    N, C = x.shape[0], x.shape[1]
    S = x.numel() // (N * C)
    out = torch.empty(N * C * S * 2, dtype=torch.uint8, device=x.device)
    e0 = _prof_begin()
    check(lib().nc_to_c8(_ptr(x), _ptr(out), I(N), I(C), L_(S), I(dt), _stream()), 'nc_to_c8')
    if e0 is not None:
        _prof_end(e0, 'to_c8', 0.0)
    return out


def _lp_ws(dims, K, k3, stride, pad, device, tag='ws_lp'):
    N, C, D, H, W = dims
    nb = lib().nc_conv_lp_ws_bytes(I(N), I(C), I(D), I(H), I(W), I(K), I(k3[0]), I(k3[1]), I(k3[2]), I(stride), I(pad))
    return workspace(nb, device, tag)


def conv_fwd_raw(x, w, b, stride, pad, xh=None):
    """xh: x already in the C8 layout (to_c8) for the 16-bit kernel of this call -- saves the conversion."""
    _chk(x, w, b)
    _f32(x, w, b)
    dims = _dims5(x.shape)
    N, C, D, H, W = dims
    k3 = _kdims(w.shape)
    K = w.shape[0]
    if w.shape[1] != C:
        raise _lib.NcError('conv: weight expects %d input channels, got %d' % (w.shape[1], C))
    y = torch.empty(_conv_out_shape(x.shape, w.shape, stride, pad), dtype=torch.float32, device=x.device)
    dt = _lp(0, dims, K, k3, stride, pad)
    if dt:
        ws = _lp_ws(dims, K, k3, stride, pad, x.device)
        e0 = _prof_begin(2.0 * C * K * k3[0] * k3[1] * k3[2] * (y.numel() // K))
        check(lib().nc_conv_fwd_lp(_ptr(x), _ptr(xh), _ptr(w), _ptr(b), _ptr(y), I(N), I(C), I(D), I(H), I(W), I(K), I(k3[0]),
                                   I(k3[1]), I(k3[2]), I(stride), I(pad), I(dt), _ptr(ws), Z(ws.numel()), _stream()),
              'nc_conv_fwd_lp')
        if e0 is not None:
            _prof_end(e0, 'fwd_lp_k%d' % k3[1], 2.0 * C * K * k3[0] * k3[1] * k3[2] * (y.numel() // K))
        return y
    ws = _conv_ws(dims, K, k3, stride, pad, x.device)
    e0 = _prof_begin(2.0 * C * K * k3[0] * k3[1] * k3[2] * (y.numel() // K))
    check(lib().nc_conv_fwd(_ptr(x), _ptr(w), _ptr(b), _ptr(y), I(N), I(C), I(D), I(H), I(W), I(K), I(k3[0]),
                            I(k3[1]), I(k3[2]), I(stride), I(pad), _ptr(ws), Z(ws.numel()), _stream()), 'nc_conv_fwd')
    if e0 is not None:
        _prof_end(e0, *_conv_tag('fwd', C, K, k3, stride, pad, y.numel() // K))
    return y


def conv_dgrad_raw(dy, w, x_shape, stride, pad, dyh=None):
    _chk(dy, w)
    _f32(dy, w)
    dx = torch.empty(tuple(x_shape), dtype=torch.float32, device=dy.device)
    dims = _dims5(x_shape)
    N, C, D, H, W = dims
    k3 = _kdims(w.shape)
    K = w.shape[0]
    dt = _lp(1, dims, K, k3, stride, pad)
    if dt:
        ws = _lp_ws(dims, K, k3, stride, pad, dy.device)
        e0 = _prof_begin(2.0 * C * K * k3[0] * k3[1] * k3[2] * (dy.numel() // K))
        check(lib().nc_conv_dgrad_lp(_ptr(dy), _ptr(dyh), _ptr(w), _ptr(dx), I(N), I(C), I(D), I(H), I(W), I(K), I(k3[0]),
                                     I(k3[1]), I(k3[2]), I(stride), I(pad), I(dt), _ptr(ws), Z(ws.numel()), _stream()),
              'nc_conv_dgrad_lp')
        if e0 is not None:
            _prof_end(e0, 'dgrad_lp_k%d' % k3[1], 2.0 * C * K * k3[0] * k3[1] * k3[2] * (dy.numel() // K))
        return dx
    ws = _conv_ws(dims, K, k3, stride, pad, dy.device)
    e0 = _prof_begin(2.0 * C * K * k3[0] * k3[1] * k3[2] * (dy.numel() // K))
    check(lib().nc_conv_dgrad(_ptr(dy), _ptr(w), _ptr(dx), I(N), I(C), I(D), I(H), I(W), I(K), I(k3[0]), I(k3[1]),
                              I(k3[2]), I(stride), I(pad), _ptr(ws), Z(ws.numel()), _stream()), 'nc_conv_dgrad')
    if e0 is not None:
        _prof_end(e0, *_conv_tag('dgrad', C, K, k3, stride, pad, dy.numel() // K))
    return dx


def conv_wgrad_raw(x, dy, w_shape, stride, pad, want_bias, ws_tag='ws', xh=None, dyh=None, x_shape=None):
    """x may be None when xh (its C8 copy) and x_shape are given: the 16-bit weight gradient never reads the fp32 x."""
    _chk(x, dy)
    _f32(x, dy)
    dims = _dims5(x.shape if x is not None else x_shape)
    N, C, D, H, W = dims
    K = w_shape[0]
    k3 = _kdims(w_shape)
    dw = torch.empty(tuple(w_shape), dtype=torch.float32, device=dy.device)
    db = torch.empty(K, dtype=torch.float32, device=dy.device) if want_bias else None
    dt = _lp(2, dims, K, k3, stride, pad)
    if x is None and not dt:
        raise _lib.NcError('conv_wgrad: the fp32 x is needed for the fp32 kernels')
    if dt:
        ws = _lp_ws(dims, K, k3, stride, pad, dy.device, ws_tag + '_lp')
        e0 = _prof_begin(2.0 * C * K * k3[0] * k3[1] * k3[2] * (dy.numel() // K))
        check(lib().nc_conv_wgrad_lp(_ptr(x), _ptr(xh), _ptr(dy), _ptr(dyh), _ptr(dw), _ptr(db), I(N), I(C), I(D), I(H), I(W), I(K), I(k3[0]),
                                     I(k3[1]), I(k3[2]), I(stride), I(pad), I(dt), _ptr(ws), Z(ws.numel()), _stream()),
              'nc_conv_wgrad_lp')
        if e0 is not None:
            _prof_end(e0, 'wgrad_lp_k%d' % k3[1], 2.0 * C * K * k3[0] * k3[1] * k3[2] * (dy.numel() // K))
        return dw, db
    ws = _conv_ws(dims, K, k3, stride, pad, x.device, ws_tag)
    e0 = _prof_begin(2.0 * C * K * k3[0] * k3[1] * k3[2] * (dy.numel() // K))
    check(lib().nc_conv_wgrad(_ptr(x), _ptr(dy), _ptr(dw), _ptr(db), I(N), I(C), I(D), I(H), I(W), I(K), I(k3[0]),
                              I(k3[1]), I(k3[2]), I(stride), I(pad), _ptr(ws), Z(ws.numel()), _stream()),
          'nc_conv_wgrad')
    if e0 is not None:
        _prof_end(e0, *_conv_tag('wgrad', C, K, k3, stride, pad, dy.numel() // K))
    return dw, db


class BiasLink:
    """Shared by a convolution and the InstanceNorm+activation right behind it (networks.py:420-423) for one forward pass.
    The gradient at the convolution's output is the dx of the norm's backward, so the convolution's bias gradient -- the
    per-channel sum of that dx -- is taken inside the norm's backward kernel while dx is in registers
    (nc_instnorm_act_bwd_dbias) and handed over here; the convolution's backward then skips its own pass over dy.
    The same object also carries the 16-bit operand copies between the two (conv_h.hip's C8 layout), in both directions:
    * forward, norm -> NEXT convolution: `want_xh` (dtype code, set by the caller that knows the next layer) makes the
      norm's forward emit its result in C8 as well (`xh`), and that convolution skips its own conversion pass;
    * backward, norm -> the convolution in FRONT of it: `want_dyh` (set by that convolution's forward) makes the norm's
      backward emit dx in C8 (`dyh`) for the data / weight gradient kernels."""
    __slots__ = ('want', 'dbias', 'want_xh', 'xh', 'xh_dt', 'want_dyh', 'dyh', 'dyh_dt')

    def __init__(self):
        self.want = False
        self.dbias = None
        self.want_xh = 0
        self.xh = None
        self.xh_dt = 0
        self.want_dyh = 0
        self.dyh = None
        self.dyh_dt = 0


def lp_fwd_dtype(x_shape, w_shape, stride, pad):
    """dtype code of the 16-bit forward kernel a convolution with this weight would use on an input of x_shape, or 0."""
    if len(x_shape) != len(w_shape) or x_shape[1] != w_shape[1]:
        return 0
    return _lp(0, _dims5(x_shape), w_shape[0], _kdims(w_shape), int(stride), int(pad))


class _Conv(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, stride, pad, link=None, prev=None):
        x = x.contiguous()
        ctx.link = link
        if link is not None:
            link.want = b is not None and ctx.needs_input_grad[2]
        ctx.cfg = (stride, pad, b is not None)
        ctx.x_shape = tuple(x.shape)
        dims, K, k3 = _dims5(x.shape), w.shape[0], _kdims(w.shape)
        dt = _lp(0, dims, K, k3, stride, pad) if w.shape[1] == x.shape[1] else 0
        if dt:
            # 16-bit path: x is converted ONCE; if the weight gradient runs in the same type it reuses that copy, and the
            # copy (half the bytes) is what is kept for backward instead of the fp32 activation
            if prev is not None and prev.xh is not None and prev.xh_dt == dt:
                xh, prev.xh = prev.xh, None  # emitted by the norm in front of this layer
            else:
                xh = to_c8(x, dt)
            y = conv_fwd_raw(x, w, b, stride, pad, xh=xh)
            keep = _lp(2, dims, K, k3, stride, pad) == dt
            if link is not None:  # backward operands: ask the norm behind this layer for dy in C8
                dd = _lp(1, dims, K, k3, stride, pad) if ctx.needs_input_grad[0] else 0
                dw_ = _lp(2, dims, K, k3, stride, pad) if (ctx.needs_input_grad[1] or link.want) else 0
                if (dd or dw_) and (not dd or not dw_ or dd == dw_):
                    link.want_dyh = dd or dw_
            ctx.x_is_c8 = keep
            ctx.save_for_backward(xh if keep else x, w)
            return y
        ctx.x_is_c8 = False
        ctx.save_for_backward(x, w)
        return conv_fwd_raw(x, w, b, stride, pad)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        stride, pad, has_b = ctx.cfg
        dy = dy.contiguous()
        dx = dw = db = None
        want_b = has_b and ctx.needs_input_grad[2]
        want_w = ctx.needs_input_grad[1] or want_b
        db_link = None
        if want_b and ctx.link is not None and ctx.link.dbias is not None:
            db_link, ctx.link.dbias = ctx.link.dbias, None  # taken by the norm's backward: no pass over dy for it here
            want_b = False
            want_w = ctx.needs_input_grad[1]
        dims, K, k3 = _dims5(ctx.x_shape), w.shape[0], _kdims(w.shape)
        dt_d = _lp(1, dims, K, k3, stride, pad) if ctx.needs_input_grad[0] else 0
        dt_w = _lp(2, dims, K, k3, stride, pad) if want_w else 0
        if dt_d or dt_w:  # 16-bit backward: dy is converted once for the data and the weight gradient
            lk = ctx.link
            if lk is not None and lk.dyh is not None and lk.dyh_dt == (dt_d or dt_w):
                dyh, lk.dyh = lk.dyh, None  # emitted by the backward of the norm behind this layer
            else:
                dyh = to_c8(dy, dt_d or dt_w)
            if ctx.needs_input_grad[0]:
                dx = conv_dgrad_raw(dy, w, ctx.x_shape, stride, pad, dyh=dyh if dt_d else None)
            if want_w:
                same = dt_w and (not dt_d or dt_d == dt_w)
                if ctx.x_is_c8:
                    dw, db = conv_wgrad_raw(None, dy, w.shape, stride, pad, want_b, xh=x, dyh=dyh if same else None,
                                            x_shape=ctx.x_shape)
                else:
                    dw, db = conv_wgrad_raw(x, dy, w.shape, stride, pad, want_b, dyh=dyh if same else None)
            return dx, dw, db if db_link is None else db_link, None, None, None, None
        big = x.numel() >= (1 << 20)  # small layers gain nothing from a second stream
        if want_w and ctx.needs_input_grad[0] and overlap_wgrad and big and prof is None:
            main = torch.cuda.current_stream()
            side = _side_stream(x.device)
            side.wait_stream(main)  # dy (and x) were produced on the main stream
            with torch.cuda.stream(side):
                dw, db = conv_wgrad_raw(x, dy, w.shape, stride, pad, want_b, 'ws_side')
            dy.record_stream(side)
            x.record_stream(side)
            dx = conv_dgrad_raw(dy, w, x.shape, stride, pad)
            main.wait_stream(side)  # dw / db are consumed (accumulated into .grad) on the main stream
            return dx, dw, db if db_link is None else db_link, None, None, None, None
        if ctx.needs_input_grad[0]:
            dx = conv_dgrad_raw(dy, w, x.shape, stride, pad)
        if want_w:
            dw, db = conv_wgrad_raw(x, dy, w.shape, stride, pad, want_b)
        return dx, dw, db if db_link is None else db_link, None, None, None, None


def conv(x, w, b=None, stride=1, padding=0, link=None, prev=None):
    """nn.Conv3d / nn.Conv2d (models/networks.py:361-369).  link / prev: BiasLink shared with the InstanceNorm behind /
    in front of this layer."""
    return _Conv.apply(x, w, b, int(stride), int(padding), link, prev)


class _ConvT(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b):
        x = x.contiguous()
        _chk(x, w, b)
        _f32(x, w, b)
        if x.dim() != 5 or tuple(w.shape[2:]) != (2, 2, 2):
            raise _lib.NcError('convT_k2s2: only ConvTranspose3d(kernel 2, stride 2) is on the hot path')
        N, C, D, H, W = x.shape
        K = w.shape[1]
        y = torch.empty((N, K, 2 * D, 2 * H, 2 * W), dtype=torch.float32, device=x.device)
        if lib().nc_convT_k2s2_split_active(I(1), I(C), I(D), I(H), I(W), I(K)):
            # the split-operand kernel of csrc/convt_s3.hip, sample by sample: exactly what the whole-network calls do
            ws = workspace(lib().nc_convT_k2s2_split_ws_bytes(I(1), I(C), I(D), I(H), I(W), I(K)), x.device, 'ws_convT_split')
            for n in range(N):
                check(lib().nc_convT_k2s2_fwd_split(_ptr(x[n]), None, _ptr(w), _ptr(b), _ptr(y[n]), None, I(0), I(0), I(1), I(C), I(D), I(H), I(W),
                                                    I(K), _ptr(ws), Z(ws.numel()), _stream()), 'nc_convT_k2s2_fwd_split')
        else:
            check(lib().nc_convT_k2s2_fwd(_ptr(x), _ptr(w), _ptr(b), _ptr(y), I(N), I(C), I(D), I(H), I(W), I(K),
                                          _stream()), 'nc_convT_k2s2_fwd')
        ctx.save_for_backward(x, w)
        ctx.has_b = b is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = dy.contiguous()
        N, C, D, H, W = x.shape
        K = w.shape[1]
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            nb = lib().nc_convT_ws_bytes(I(N), I(C), I(D), I(H), I(W), I(K))
            ws = workspace(nb, x.device, 'ws_convT')
            check(lib().nc_convT_k2s2_dgrad(_ptr(dy), _ptr(w), _ptr(dx), I(N), I(C), I(D), I(H), I(W), I(K), _ptr(ws),
                                            Z(ws.numel()), _stream()), 'nc_convT_k2s2_dgrad')
        if ctx.needs_input_grad[1]:
            dw = torch.empty_like(w)
            db = torch.empty(K, dtype=torch.float32, device=x.device) if ctx.has_b else None
            ws = workspace(lib().nc_convT_ws_bytes(I(N), I(C), I(D), I(H), I(W), I(K)), x.device)
            check(lib().nc_convT_k2s2_wgrad(_ptr(x), _ptr(dy), _ptr(dw), _ptr(db), I(N), I(C), I(D), I(H), I(W), I(K),
                                            _ptr(ws), Z(ws.numel()), _stream()), 'nc_convT_k2s2_wgrad')
        return dx, dw, db


def conv_transpose_k2s2(x, w, b=None):
    """nn.ConvTranspose3d(C, K, 2, 2) (models/networks.py:500,503)."""
    return _ConvT.apply(x, w, b)


def instnorm_stats(x, eps=1e-5):
    _chk(x)
    _f32(x)
    NC = x.shape[0] * x.shape[1]
    S = x.numel() // NC
    mean = torch.empty(NC, dtype=torch.float32, device=x.device)
    rstd = torch.empty(NC, dtype=torch.float32, device=x.device)
    ws = workspace(lib().nc_instnorm_ws_bytes(I(NC), L_(S)), x.device, 'in')
    check(lib().nc_instnorm_stats(_ptr(x), I(NC), L_(S), F(eps), _ptr(mean), _ptr(rstd), _ptr(ws), Z(ws.numel()),
                                  _stream()), 'nc_instnorm_stats')
    return mean, rstd


class _InstNormAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, slope, eps, link=None, nxt=None):
        x = x.contiguous()
        ctx.link = link
        y = torch.empty_like(x)
        if nxt is not None and nxt.want_xh and x.shape[1] % 8 == 0:
            mean, rstd = instnorm_stats(x, eps)
            NC = mean.numel()
            S = x.numel() // NC
            yh = torch.empty(x.numel() * 2, dtype=torch.uint8, device=x.device)
            check(lib().nc_instnorm_act_fwd_c8(_ptr(x), _ptr(mean), _ptr(rstd), F(slope), _ptr(y), _ptr(yh), I(x.shape[0]),
                                               I(x.shape[1]), L_(S), I(nxt.want_xh), _stream()), 'nc_instnorm_act_fwd_c8')
            nxt.xh, nxt.xh_dt = yh, nxt.want_xh
        else:
            _chk(x)
            _f32(x)
            NC = x.shape[0] * x.shape[1]
            S = x.numel() // NC
            mean = torch.empty(NC, dtype=torch.float32, device=x.device)
            rstd = torch.empty(NC, dtype=torch.float32, device=x.device)
            ws = workspace(lib().nc_instnorm_ws_bytes(I(NC), L_(S)), x.device, 'in')
            check(lib().nc_instnorm_fwd(_ptr(x), F(eps), F(slope), _ptr(mean), _ptr(rstd), _ptr(y), I(NC), L_(S), _ptr(ws),
                                        Z(ws.numel()), _stream()), 'nc_instnorm_fwd')
        ctx.save_for_backward(x, mean, rstd)
        ctx.slope = slope
        return y

    @staticmethod
    def backward(ctx, dy):
        x, mean, rstd = ctx.saved_tensors
        dy = dy.contiguous()
        NC = mean.numel()
        S = x.numel() // NC
        dx = torch.empty_like(x)
        link = ctx.link
        if link is not None and link.want_dyh and x.shape[1] % 8 == 0:
            N, C = x.shape[0], x.shape[1]
            db = torch.empty(C, dtype=torch.float32, device=x.device) if link.want else None
            dxh = torch.empty(x.numel() * 2, dtype=torch.uint8, device=x.device)
            ws = workspace(lib().nc_instnorm_bwd_dbias_ws_bytes(I(NC), L_(S)), x.device, 'in')
            check(lib().nc_instnorm_act_bwd_c8(_ptr(dy), _ptr(x), _ptr(mean), _ptr(rstd), F(ctx.slope), _ptr(dx), _ptr(dxh),
                                               _ptr(db), I(N), I(C), L_(S), I(link.want_dyh), _ptr(ws), Z(ws.numel()),
                                               _stream()), 'nc_instnorm_act_bwd_c8')
            link.dyh, link.dyh_dt, link.dbias = dxh, link.want_dyh, db
            return dx, None, None, None, None
        if link is not None and link.want:
            N, C = x.shape[0], x.shape[1]
            db = torch.empty(C, dtype=torch.float32, device=x.device)
            ws = workspace(lib().nc_instnorm_bwd_dbias_ws_bytes(I(NC), L_(S)), x.device, 'in')
            check(lib().nc_instnorm_act_bwd_dbias(_ptr(dy), _ptr(x), _ptr(mean), _ptr(rstd), F(ctx.slope), _ptr(dx), _ptr(db),
                                                  I(N), I(C), L_(S), _ptr(ws), Z(ws.numel()), _stream()),
                  'nc_instnorm_act_bwd_dbias')
            link.dbias = db
            return dx, None, None, None, None
        ws = workspace(lib().nc_instnorm_ws_bytes(I(NC), L_(S)), x.device, 'in')
        check(lib().nc_instnorm_act_bwd(_ptr(dy), _ptr(x), _ptr(mean), _ptr(rstd), F(ctx.slope), _ptr(dx), I(NC),
                                        L_(S), _ptr(ws), Z(ws.numel()), _stream()), 'nc_instnorm_act_bwd')
        return dx, None, None, None, None


def instance_norm_act(x, slope=0.0, eps=1e-5, link=None, nxt=None):
    """InstanceNorm{2,3}d(affine=False) followed by ReLU (slope 0) / LeakyReLU(slope) -- networks.py:33-34,422-423.
    link / nxt: BiasLink shared with the convolution in front of / behind this layer."""
    return _InstNormAct.apply(x, float(slope), float(eps), link, nxt)


class _BatchNormAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, training, momentum, eps, slope):
        x = x.contiguous()
        _chk(x, gamma, beta)
        _f32(x, gamma, beta)
        N, C = x.shape[0], x.shape[1]
        S = x.numel() // (N * C)
        L = lib()
        mean = torch.empty(C, dtype=torch.float32, device=x.device)
        rstd = torch.empty(C, dtype=torch.float32, device=x.device)
        ws = workspace(L.nc_instnorm_ws_bytes(I(N * C), L_(S)), x.device, 'bn')
        check(L.nc_batchnorm_stats(_ptr(x), I(N), I(C), L_(S), F(eps), F(momentum), I(1 if training else 0), _ptr(mean), _ptr(rstd),
                                   _ptr(running_mean), _ptr(running_var), _ptr(ws), Z(ws.numel()), _stream()), 'nc_batchnorm_stats')
        y = torch.empty_like(x)
        check(L.nc_batchnorm_act_fwd(_ptr(x), _ptr(mean), _ptr(rstd), _ptr(gamma), _ptr(beta), F(slope), _ptr(y), I(N), I(C), L_(S), _stream()),
              'nc_batchnorm_act_fwd')
        ctx.save_for_backward(x, mean, rstd, gamma, beta)
        ctx.cfg = (bool(training), float(slope))
        return y

    @staticmethod
    def backward(ctx, dy):
        x, mean, rstd, gamma, beta = ctx.saved_tensors
        training, slope = ctx.cfg
        dy = dy.contiguous()
        N, C = x.shape[0], x.shape[1]
        S = x.numel() // (N * C)
        L = lib()
        dx = torch.empty_like(x)
        dgamma, dbeta = torch.empty_like(gamma), torch.empty_like(beta)
        coef = torch.empty(2 * C, dtype=torch.float32, device=x.device)
        ws = workspace(L.nc_instnorm_ws_bytes(I(N * C), L_(S)), x.device, 'bn')
        check(L.nc_batchnorm_act_bwd(_ptr(dy), _ptr(x), _ptr(mean), _ptr(rstd), _ptr(gamma), _ptr(beta), F(slope), I(1 if training else 0),
                                     _ptr(dx), _ptr(dgamma), _ptr(dbeta), _ptr(coef), I(N), I(C), L_(S), _ptr(ws), Z(ws.numel()), _stream()),
              'nc_batchnorm_act_bwd')
        return dx, dgamma, dbeta, None, None, None, None, None, None


def batch_norm_act(x, gamma, beta, running_mean, running_var, training, momentum=0.1, eps=1e-5, slope=0.0):
    """nn.BatchNorm{2,3}d(affine=True, track_running_stats=True) (models/networks.py:30-31, --norm batch) followed by ReLU (slope 0) /
    LeakyReLU(slope); running_mean / running_var are updated in place in training mode and used in evaluation mode."""
    return _BatchNormAct.apply(x, gamma, beta, running_mean, running_var, bool(training), float(momentum), float(eps), float(slope))


class _LeakyReLU(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, slope):
        x = x.contiguous()
        _chk(x)
        _f32(x)
        y = torch.empty_like(x)
        check(lib().nc_leaky_relu_fwd(_ptr(x), F(slope), _ptr(y), L_(x.numel()), _stream()), 'nc_leaky_relu_fwd')
        ctx.save_for_backward(x)
        ctx.slope = slope
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        check(lib().nc_leaky_relu_bwd(_ptr(dy), _ptr(x), F(ctx.slope), _ptr(dx), L_(x.numel()), _stream()),
              'nc_leaky_relu_bwd')
        return dx, None


def leaky_relu(x, slope):
    return _LeakyReLU.apply(x, float(slope))


class _Sigmoid(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        _chk(x)
        _f32(x)
        y = torch.empty_like(x)
        check(lib().nc_sigmoid_fwd(_ptr(x), _ptr(y), L_(x.numel()), _stream()), 'nc_sigmoid_fwd')
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        dy = dy.contiguous()
        dx = torch.empty_like(y)
        check(lib().nc_sigmoid_bwd(_ptr(dy), _ptr(y), _ptr(dx), L_(y.numel()), _stream()), 'nc_sigmoid_bwd')
        return dx


def sigmoid(x):
    return _Sigmoid.apply(x)


class _MaxPool2(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        _chk(x)
        _f32(x)
        N, C, D, H, W = _dims5(x.shape)
        wd = 2 if D > 1 else 1
        oshape = (N, C, D // wd, H // 2, W // 2) if x.dim() == 5 else (N, C, H // 2, W // 2)
        y = torch.empty(oshape, dtype=torch.float32, device=x.device)
        check(lib().nc_maxpool2_fwd(_ptr(x), _ptr(y), I(N * C), I(D), I(H), I(W), _stream()), 'nc_maxpool2_fwd')
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        dy = dy.contiguous()
        N, C, D, H, W = _dims5(x.shape)
        dx = torch.empty_like(x)
        check(lib().nc_maxpool2_bwd(_ptr(dy), _ptr(x), _ptr(dx), I(N * C), I(D), I(H), I(W), _stream()),
              'nc_maxpool2_bwd')
        return dx


def maxpool2(x):
    """nn.MaxPool3d(2) (models/networks.py:491,494)."""
    return _MaxPool2.apply(x)


_PLANE = [lambda N, C, D, H, W: (N, C, H, W), lambda N, C, D, H, W: (N, C, D, W), lambda N, C, D, H, W: (N, C, D, H)]


class _Slice(torch.autograd.Function):
    @staticmethod
    def forward(ctx, vol, axis, index):
        vol = vol.contiguous()
        _chk(vol)
        _f32(vol)
        N, C, D, H, W = vol.shape
        out = torch.empty(_PLANE[axis](N, C, D, H, W), dtype=torch.float32, device=vol.device)
        check(lib().nc_slice_fwd(_ptr(vol), _ptr(out), I(N * C), I(D), I(H), I(W), I(axis), I(index), _stream()),
              'nc_slice_fwd')
        ctx.cfg = (tuple(vol.shape), axis, index)
        return out

    @staticmethod
    def backward(ctx, dout):
        shape, axis, index = ctx.cfg
        dout = dout.contiguous()
        N, C, D, H, W = shape
        dvol = torch.empty(shape, dtype=torch.float32, device=dout.device)
        check(lib().nc_slice_bwd(_ptr(dout), _ptr(dvol), I(N * C), I(D), I(H), I(W), I(axis), I(index), _stream()),
              'nc_slice_bwd')
        return dvol, None, None


def volume_slice(vol, axis, index):
    """Volume.get_slice (apollo_model.py:328-337)."""
    return _Slice.apply(vol, int(axis), int(index))


class _Mip(torch.autograd.Function):
    @staticmethod
    def forward(ctx, vol, axis, start, depth):
        vol = vol.contiguous()
        _chk(vol)
        _f32(vol)
        N, C, D, H, W = vol.shape
        oshape = _PLANE[axis](N, C, D, H, W)
        out = torch.empty(oshape, dtype=torch.float32, device=vol.device)
        arg = torch.empty(oshape, dtype=torch.int32, device=vol.device)
        check(lib().nc_mip_fwd(_ptr(vol), _ptr(out), _ptr(arg), I(N * C), I(D), I(H), I(W), I(axis), I(start),
                               I(depth), _stream()), 'nc_mip_fwd')
        ctx.save_for_backward(arg)
        ctx.cfg = (tuple(vol.shape), axis)
        return out

    @staticmethod
    def backward(ctx, dout):
        (arg,) = ctx.saved_tensors
        shape, axis = ctx.cfg
        dout = dout.contiguous()
        N, C, D, H, W = shape
        dvol = torch.empty(shape, dtype=torch.float32, device=dout.device)
        check(lib().nc_mip_bwd(_ptr(dout), _ptr(arg), _ptr(dvol), I(N * C), I(D), I(H), I(W), I(axis), _stream()),
              'nc_mip_bwd')
        return dvol, None, None, None


def volume_mip(vol, axis, start, depth):
    """Volume.get_projection (apollo_model.py:339-351)."""
    return _Mip.apply(vol, int(axis), int(start), int(depth))


class _AllSlices(torch.autograd.Function):
    @staticmethod
    def forward(ctx, vol, axis):
        vol = vol.contiguous()
        _chk(vol)
        _f32(vol)
        N, C, D, H, W = vol.shape
        L = (D, H, W)[axis]
        pn, pc, pa, pb = _PLANE[axis](N, C, D, H, W)
        out = torch.empty((N * L, C, pa, pb), dtype=torch.float32, device=vol.device)
        check(lib().nc_volume_slices(_ptr(vol), _ptr(out), I(N), I(C), I(D), I(H), I(W), I(axis), I(0), _stream()),
              'nc_volume_slices')
        ctx.cfg = (tuple(vol.shape), axis)
        return out

    @staticmethod
    def backward(ctx, dout):
        shape, axis = ctx.cfg
        dout = dout.contiguous()
        N, C, D, H, W = shape
        dvol = torch.empty(shape, dtype=torch.float32, device=dout.device)
        check(lib().nc_volume_slices(_ptr(dout), _ptr(dvol), I(N), I(C), I(D), I(H), I(W), I(axis), I(1), _stream()),
              'nc_volume_slices')
        return dvol, None


def volume_all_slices(vol, axis):
    """Every slice along `axis` as a batch [(n, s), C, A, B] -- the batched form of Athena's iter_f loop
    (axial_to_lateral_gan_athena_model.py:286-296)."""
    return _AllSlices.apply(vol, int(axis))


class _MseConst(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target):
        pred = pred.contiguous()
        _chk(pred)
        _f32(pred)
        out = torch.empty((), dtype=torch.float32, device=pred.device)
        ws = workspace(lib().nc_loss_ws_bytes(L_(pred.numel())), pred.device, 'loss')
        check(lib().nc_mse_const_fwd(_ptr(pred), L_(pred.numel()), F(target), _ptr(out), _ptr(ws), Z(ws.numel()),
                                     _stream()), 'nc_mse_const_fwd')
        ctx.save_for_backward(pred)
        ctx.target = target
        return out

    @staticmethod
    def backward(ctx, g):
        (pred,) = ctx.saved_tensors
        g = g.contiguous()
        dp = torch.empty_like(pred)
        check(lib().nc_mse_const_bwd(_ptr(pred), L_(pred.numel()), F(ctx.target), _ptr(g), _ptr(dp), _stream()),
              'nc_mse_const_bwd')
        return dp, None


def mse_const(pred, target):
    """GANLoss('lsgan') (models/networks.py:276,299-313): mean((pred - target)^2), target a constant."""
    return _MseConst.apply(pred, float(target))


class _BceLogitsConst(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target):
        pred = pred.contiguous()
        _chk(pred)
        _f32(pred)
        out = torch.empty((), dtype=torch.float32, device=pred.device)
        ws = workspace(lib().nc_loss_ws_bytes(L_(pred.numel())), pred.device, 'loss')
        check(lib().nc_bce_logits_const_fwd(_ptr(pred), L_(pred.numel()), F(target), _ptr(out), _ptr(ws), Z(ws.numel()), _stream()),
              'nc_bce_logits_const_fwd')
        ctx.save_for_backward(pred)
        ctx.target = target
        return out

    @staticmethod
    def backward(ctx, g):
        (pred,) = ctx.saved_tensors
        g = g.contiguous()
        dp = torch.empty_like(pred)
        check(lib().nc_bce_logits_const_bwd(_ptr(pred), L_(pred.numel()), F(ctx.target), _ptr(g), _ptr(dp), _stream()),
              'nc_bce_logits_const_bwd')
        return dp, None


def bce_logits_const(pred, target):
    """GANLoss('vanilla') (models/networks.py:278, 308-313): nn.BCEWithLogitsLoss of the logits against a constant label."""
    return _BceLogitsConst.apply(pred, float(target))


class _Mean(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred):
        pred = pred.contiguous()
        _chk(pred)
        _f32(pred)
        out = torch.empty((), dtype=torch.float32, device=pred.device)
        ws = workspace(lib().nc_loss_ws_bytes(L_(pred.numel())), pred.device, 'loss')
        check(lib().nc_mean_fwd(_ptr(pred), L_(pred.numel()), _ptr(out), _ptr(ws), Z(ws.numel()), _stream()), 'nc_mean_fwd')
        ctx.shape = pred.shape
        return out

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        dp = torch.empty(ctx.shape, dtype=torch.float32, device=g.device)
        check(lib().nc_mean_bwd(L_(dp.numel()), _ptr(g), _ptr(dp), _stream()), 'nc_mean_bwd')
        return dp


def mean(pred):
    """prediction.mean() of GANLoss('wgangp') (models/networks.py:314-318); the caller negates it for real targets."""
    return _Mean.apply(pred)


class _L1(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        a, b = a.contiguous(), b.contiguous()
        _chk(a, b)
        _f32(a, b)
        out = torch.empty((), dtype=torch.float32, device=a.device)
        ws = workspace(lib().nc_loss_ws_bytes(L_(a.numel())), a.device, 'loss')
        check(lib().nc_l1_fwd(_ptr(a), _ptr(b), L_(a.numel()), _ptr(out), _ptr(ws), Z(ws.numel()), _stream()),
              'nc_l1_fwd')
        ctx.save_for_backward(a, b)
        return out

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        g = g.contiguous()
        da = torch.empty_like(a)
        check(lib().nc_l1_bwd(_ptr(a), _ptr(b), L_(a.numel()), _ptr(g), _ptr(da), _stream()), 'nc_l1_bwd')
        return da, None


def l1_loss(a, b):
    """torch.nn.L1Loss()(a, b) with b treated as a constant (apollo_model.py:128,279: rec vs real)."""
    return _L1.apply(a, b)


def adam_step(p, g, m, v, lr, beta1, beta2, eps, step):
    """torch.optim.Adam.step over one flat buffer (apollo_model.py:131-136)."""
    _chk(p, g, m, v)
    _f32(p, g, m, v)
    check(lib().nc_adam_step(_ptr(p), _ptr(g), _ptr(m), _ptr(v), L_(p.numel()), F(lr), F(beta1), F(beta2), F(eps),
                             I(step), _stream()), 'nc_adam_step')


def set_force_direct(on):
    lib().nc_set_force_direct(I(1 if on else 0))


# Generation counter per parameter storage: nc_adam_step (and checkpoint loads) write parameters through raw pointers,
# which autograd's version counters never see.  FlatAdam.step / BaseModel.load_networks bump it; the whole-network
# Functions, which keep a zero-copy alias of the live parameters for backward, refuse a backward across a bump.
_param_gen = {}


def bump_param_generation(t):
    k = t.untyped_storage().data_ptr()
    _param_gen[k] = _param_gen.get(k, 0) + 1


def _param_generation(t):
    return _param_gen.get(t.untyped_storage().data_ptr(), 0)


# Direct gradient destination of the whole-network calls.  FlatAdam registers its flat gradient buffer under its flat
# parameter buffer's storage; when a whole-network backward finds that its packed parameters ARE a slice of such a
# buffer, it lets the kernels write the parameter gradients straight into the matching slice of the gradient buffer
# (they overwrite, so only the first backward of a slice per optimizer step may do that) and returns views of it: with
# p.grad = None autograd's AccumulateGrad adopts them -- no per-parameter add kernels, no extra copy.
_flat_grads = {}   # storage ptr of the flat parameter buffer -> [grad buffer, set of slices written this step]


def register_flat_grad(flat, grad):
    _flat_grads[flat.untyped_storage().data_ptr()] = [grad, set()]


def flat_grad_step_begin(flat):
    ent = _flat_grads.get(flat.untyped_storage().data_ptr())
    if ent is not None:
        ent[1].clear()


def _grad_destination(packed):
    """A slice of the registered flat gradient buffer for these packed parameters, or a fresh tensor."""
    ent = _flat_grads.get(packed.untyped_storage().data_ptr())
    if ent is not None and os.environ.get('NC_DIRECT_GRADS', '1') != '0':
        grad, written = ent
        off, n = packed.storage_offset(), packed.numel()
        if off + n <= grad.numel() and not any(a < off + n and off < b for a, b in written):
            written.add((off, off + n))
            return grad[off:off + n]
    return torch.empty_like(packed)


# ---- whole-network PatchGAN (nc_patchgan_fwd / nc_patchgan_bwd): one C call per direction -------------------------
def _pack_params(params):
    """The parameter tensors as ONE flat fp32 tensor in the given order: a zero-copy view when they already sit back
    to back in one storage (FlatAdam's flat buffer), otherwise a concatenated copy."""
    p0 = params[0]
    off = p0.storage_offset()
    same = True
    for p in params:
        if (not p.is_contiguous()) or p.untyped_storage().data_ptr() != p0.untyped_storage().data_ptr() or \
                p.storage_offset() != off:
            same = False
            break
        off += p.numel()
    total = sum(p.numel() for p in params)
    if same:
        return p0.detach().as_strided((total,), (1,), p0.storage_offset())
    return torch.cat([p.detach().reshape(-1) for p in params])


class _PatchGAN(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, cfg, *params):
        n_layers, ndf, nd = cfg
        x = x.contiguous()
        _chk(x, *params)
        _f32(x, *params)
        if x.shape[1] != 1:
            raise _lib.NcError('fused PatchGAN expects one input channel')
        B = x.shape[0]
        D, H, W = (1, x.shape[2], x.shape[3]) if nd == 2 else tuple(x.shape[2:])
        L = lib()
        packed = _pack_params(params)
        if packed.numel() != L.nc_patchgan_param_floats(I(n_layers), I(ndf), I(nd)):
            raise _lib.NcError('fused PatchGAN: parameter count does not match (n_layers=%d, ndf=%d)' % (n_layers, ndf))
        od, oh, ow = I(0), I(0), I(0)
        import ctypes
        check(L.nc_patchgan_out_shape(I(B), I(D), I(H), I(W), I(n_layers), I(ndf), I(nd), ctypes.byref(od),
                                      ctypes.byref(oh), ctypes.byref(ow)), 'nc_patchgan_out_shape')
        oshape = (B, 1, oh.value, ow.value) if nd == 2 else (B, 1, od.value, oh.value, ow.value)
        y = torch.empty(oshape, dtype=torch.float32, device=x.device)
        saved = torch.empty(L.nc_patchgan_saved_floats(I(B), I(D), I(H), I(W), I(n_layers), I(ndf), I(nd)),
                            dtype=torch.float32, device=x.device)
        ws = workspace(L.nc_patchgan_ws_bytes(I(B), I(D), I(H), I(W), I(n_layers), I(ndf), I(nd)), x.device, 'patchgan')
        check(L.nc_patchgan_fwd(_ptr(packed), _ptr(x), _ptr(y), _ptr(saved), I(B), I(D), I(H), I(W), I(n_layers), I(ndf),
                                I(nd), _ptr(ws), Z(ws.numel()), _stream()), 'nc_patchgan_fwd')
        ctx.save_for_backward(x, saved)
        ctx.packed = packed
        ctx.packed_gen = _param_generation(packed)
        ctx.cfg = (cfg, (B, D, H, W), [tuple(p.shape) for p in params])
        return y

    @staticmethod
    def backward(ctx, dy):
        x, saved = ctx.saved_tensors
        (n_layers, ndf, nd), (B, D, H, W), shapes = ctx.cfg
        if _param_generation(ctx.packed) != ctx.packed_gen:
            raise _lib.NcError('fused PatchGAN: the parameters were updated (optimizer step / checkpoint load) between '
                               'this forward and its backward; the saved activations no longer match them')
        dy = dy.contiguous()
        want_x = ctx.needs_input_grad[0]
        want_p = any(ctx.needs_input_grad[2:])
        dx = torch.empty_like(x) if want_x else None
        dpar = _grad_target(ctx, 2) if want_p else None  # (x, cfg, *params): the parameters start at input 2
        L = lib()
        ws = workspace(L.nc_patchgan_ws_bytes(I(B), I(D), I(H), I(W), I(n_layers), I(ndf), I(nd)), x.device, 'patchgan')
        check(L.nc_patchgan_bwd(_ptr(ctx.packed), _ptr(x), _ptr(saved), _ptr(dy), _ptr(dx), _ptr(dpar), I(B), I(D), I(H),
                                I(W), I(n_layers), I(ndf), I(nd), _ptr(ws), Z(ws.numel()), _stream()), 'nc_patchgan_bwd')
        grads = [None] * len(shapes)
        if want_p:
            off = 0
            for i, shp in enumerate(shapes):
                n = 1
                for s in shp:
                    n *= s
                if ctx.needs_input_grad[2 + i]:
                    grads[i] = dpar[off:off + n].view(shp)
                off += n
        return (dx, None) + tuple(grads)


class PatchGANShare:
    """One discriminator, one optimisation step: the activations of the pass over the slices of `fake` in the generator
    loss, laid out as the SECOND half of the (real, fake) batch of the discriminator loss that follows with the same
    weights (athena_model.py:240-260 then :190-238: optimizer_D.step() comes after both).  patchgan_fake_half fills it,
    patchgan_join_real runs only the `real` half and back-propagates through the whole batch."""
    __slots__ = ('saved', 'x', 'y', 'gen', 'packed_ptr', 'cfg', 'dims', 'src', 'src_version', 'axis')

    def __init__(self):
        self.saved = None

    def matches(self, params, cfg, fake, axis):
        if self.saved is None or self.cfg != tuple(cfg) or self.src is not fake or self.src_version != fake._version \
                or self.axis != axis:
            return False
        packed = _pack_params(params)
        return packed.data_ptr() == self.packed_ptr and _param_generation(packed) == self.gen

    def release(self):
        self.saved = self.x = self.y = self.src = None


def _pg_dims(x, nd):
    return (1, x.shape[2], x.shape[3]) if nd == 2 else tuple(x.shape[2:])


class _PatchGANFakeHalf(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, cfg, share, *params):
        n_layers, ndf, nd = cfg
        x = x.contiguous()
        _chk(x, *params)
        _f32(x, *params)
        if x.shape[1] != 1:
            raise _lib.NcError('fused PatchGAN expects one input channel')
        B = x.shape[0]
        D, H, W = _pg_dims(x, nd)
        L = lib()
        packed = _pack_params(params)
        if packed.numel() != L.nc_patchgan_param_floats(I(n_layers), I(ndf), I(nd)):
            raise _lib.NcError('fused PatchGAN: parameter count does not match (n_layers=%d, ndf=%d)' % (n_layers, ndf))
        import ctypes
        od, oh, ow = I(0), I(0), I(0)
        check(L.nc_patchgan_out_shape(I(2 * B), I(D), I(H), I(W), I(n_layers), I(ndf), I(nd), ctypes.byref(od),
                                      ctypes.byref(oh), ctypes.byref(ow)), 'nc_patchgan_out_shape')
        oshape = (2 * B, 1, oh.value, ow.value) if nd == 2 else (2 * B, 1, od.value, oh.value, ow.value)
        share.saved = torch.empty(L.nc_patchgan_saved_floats(I(2 * B), I(D), I(H), I(W), I(n_layers), I(ndf), I(nd)),
                                  dtype=torch.float32, device=x.device)
        share.x = torch.empty((2 * B,) + tuple(x.shape[1:]), dtype=torch.float32, device=x.device)
        share.x[B:].copy_(x)
        share.y = torch.empty(oshape, dtype=torch.float32, device=x.device)
        ws = workspace(L.nc_patchgan_ws_bytes(I(2 * B), I(D), I(H), I(W), I(n_layers), I(ndf), I(nd)), x.device, 'patchgan')
        check(L.nc_patchgan_fwd_part(_ptr(packed), _ptr(share.x[B:]), _ptr(share.y[B:]), _ptr(share.saved), I(2 * B), I(B), I(B),
                                     I(D), I(H), I(W), I(n_layers), I(ndf), I(nd), _ptr(ws), Z(ws.numel()), _stream()),
              'nc_patchgan_fwd_part')
        share.gen, share.packed_ptr, share.cfg, share.dims = _param_generation(packed), packed.data_ptr(), tuple(cfg), (B, D, H, W)
        ctx.share, ctx.packed, ctx.cfg = share, packed, cfg
        ctx.saved_ref = share.saved  # keeps the buffers alive for this backward even if the share is released first
        ctx.x_ref = share.x
        return share.y[B:].clone()

    @staticmethod
    def backward(ctx, dy):
        n_layers, ndf, nd = ctx.cfg
        share = ctx.share
        B, D, H, W = share.dims
        if _param_generation(ctx.packed) != share.gen:
            raise _lib.NcError('fused PatchGAN: the parameters were updated between this forward and its backward')
        if any(ctx.needs_input_grad[3:]):
            raise _lib.NcError('patchgan_fake_half: the discriminator must be frozen (generator loss)')
        if not ctx.needs_input_grad[0]:
            return (None,) * (3 + len(ctx.needs_input_grad[3:]))
        dy = dy.contiguous()
        xs = ctx.x_ref[B:]
        dx = torch.empty_like(xs)
        L = lib()
        ws = workspace(L.nc_patchgan_ws_bytes(I(2 * B), I(D), I(H), I(W), I(n_layers), I(ndf), I(nd)), dy.device, 'patchgan')
        check(L.nc_patchgan_bwd_part(_ptr(ctx.packed), _ptr(xs), _ptr(ctx.saved_ref), _ptr(dy), _ptr(dx), I(2 * B), I(B), I(B),
                                     I(D), I(H), I(W), I(n_layers), I(ndf), I(nd), _ptr(ws), Z(ws.numel()), _stream()),
              'nc_patchgan_bwd_part')
        return (dx, None, None) + (None,) * len(ctx.needs_input_grad[3:])


class _PatchGANJoinReal(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, cfg, share, *params):
        n_layers, ndf, nd = cfg
        x = x.contiguous()
        _chk(x, *params)
        _f32(x, *params)
        B, D, H, W = share.dims
        if tuple(x.shape) != tuple(share.x.shape[1:]) and tuple(x.shape) != (B,) + tuple(share.x.shape[1:]):
            raise _lib.NcError('patchgan_join_real: the real planes do not have the shape of the cached fake planes')
        packed = _pack_params(params)
        if packed.data_ptr() != share.packed_ptr or _param_generation(packed) != share.gen:
            raise _lib.NcError('patchgan_join_real: the parameters changed since the pass over the fake planes')
        cur = torch.cuda.current_stream()
        for t in (share.saved, share.x, share.y):
            t.record_stream(cur)
        share.x[:B].copy_(x)
        L = lib()
        ws = workspace(L.nc_patchgan_ws_bytes(I(2 * B), I(D), I(H), I(W), I(n_layers), I(ndf), I(nd)), x.device, 'patchgan')
        check(L.nc_patchgan_fwd_part(_ptr(packed), _ptr(share.x), _ptr(share.y), _ptr(share.saved), I(2 * B), I(0), I(B), I(D),
                                     I(H), I(W), I(n_layers), I(ndf), I(nd), _ptr(ws), Z(ws.numel()), _stream()),
              'nc_patchgan_fwd_part')
        ctx.save_for_backward(share.x, share.saved)
        ctx.packed, ctx.gen, ctx.cfg, ctx.dims = packed, share.gen, cfg, share.dims
        ctx.shapes = [tuple(p.shape) for p in params]
        y = share.y
        share.release()  # the autograd graph owns the buffers from here
        return y

    @staticmethod
    def backward(ctx, dy):
        x, saved = ctx.saved_tensors
        n_layers, ndf, nd = ctx.cfg
        B, D, H, W = ctx.dims
        if _param_generation(ctx.packed) != ctx.gen:
            raise _lib.NcError('fused PatchGAN: the parameters were updated between this forward and its backward')
        dy = dy.contiguous()
        want_p = any(ctx.needs_input_grad[3:])
        dpar = _grad_target(ctx, 3) if want_p else None
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        L = lib()
        ws = workspace(L.nc_patchgan_ws_bytes(I(2 * B), I(D), I(H), I(W), I(n_layers), I(ndf), I(nd)), x.device, 'patchgan')
        check(L.nc_patchgan_bwd(_ptr(ctx.packed), _ptr(x), _ptr(saved), _ptr(dy), _ptr(dx), _ptr(dpar), I(2 * B), I(D), I(H),
                                I(W), I(n_layers), I(ndf), I(nd), _ptr(ws), Z(ws.numel()), _stream()), 'nc_patchgan_bwd')
        grads = [None] * len(ctx.shapes)
        if want_p:
            off = 0
            for i, shp in enumerate(ctx.shapes):
                n = 1
                for s_ in shp:
                    n *= s_
                if ctx.needs_input_grad[3 + i]:
                    grads[i] = dpar[off:off + n].view(shp)
                off += n
        return (dx[:B] if dx is not None else None, None, None) + tuple(grads)


def patchgan_fake_half(x, params, n_layers, ndf, dimension, share, src=None, axis=None):
    """The frozen discriminator on the slices of `fake` (generator loss), kept in `share` as the second half of the batch
    the discriminator loss runs next (see PatchGANShare).  src / axis: the volume the planes were cut from and the axis,
    which PatchGANShare.matches compares before the cached half is used."""
    y = _PatchGANFakeHalf.apply(x, (int(n_layers), int(ndf), int(dimension)), share, *params)
    share.src, share.src_version, share.axis = src, (src._version if src is not None else None), axis
    return y


def patchgan_join_real(x, params, n_layers, ndf, dimension, share):
    """Predictions for the batch (x, cached fake planes): forward of the `real` half only; backward over the whole batch."""
    return _PatchGANJoinReal.apply(x, (int(n_layers), int(ndf), int(dimension)), share, *params)


def patchgan(x, params, n_layers, ndf, dimension):
    """NLayerDiscriminator.forward (networks.py:1063-1066) with InstanceNorm, as one C call (and one for backward)."""
    return _PatchGAN.apply(x, (int(n_layers), int(ndf), int(dimension)), *params)


# ---- whole-network generators (nc_unet_deconv_train_fwd / _bwd, nc_deep_linear_fwd / _bwd): one C call per direction --
def _grad_target(ctx, first):
    """Where a whole-network backward writes its packed parameter gradients: the optimizer's flat gradient slice when EVERY parameter
    wants a gradient (the kernels OVERWRITE the whole slice), a scratch tensor otherwise -- with all parameters frozen
    (set_requires_grad(False)) nothing is handed on; with SOME frozen, autograd receives views of the scratch tensor for the others
    (FlatAdam._collect copies them into the flat buffer) and the frozen ranges of the flat gradient keep the zeros of zero_grad, so
    FlatAdam.step leaves those parameters where they are."""
    if all(ctx.needs_input_grad[first:]):
        return _grad_destination(ctx.packed)
    return torch.empty_like(ctx.packed)


def _param_grads(ctx, dpar, shapes, first):
    grads = [None] * len(shapes)
    off = 0
    for i, shp in enumerate(shapes):
        n = 1
        for s in shp:
            n *= s
        if ctx.needs_input_grad[first + i]:
            grads[i] = dpar[off:off + n].view(shp)
        off += n
    return grads


class _UnetDeconvTrain(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, *params):
        x = x.contiguous()
        _chk(x, *params)
        _f32(x, *params)
        N, _, S0, S1, S2 = x.shape
        L = lib()
        packed = _pack_params(params)
        if packed.numel() != L.nc_unet_deconv_param_floats():
            raise _lib.NcError('fused Unet_deconv: parameter count does not match')
        nsv = L.nc_unet_deconv_saved_floats(I(N), I(S0), I(S1), I(S2))
        if nsv == 0:
            raise _lib.NcError('fused Unet_deconv: every edge must be a positive multiple of 4, got %s' % ((S0, S1, S2),))
        saved = torch.empty(nsv, dtype=torch.float32, device=x.device)
        ws = workspace(L.nc_unet_deconv_train_ws_bytes(I(N), I(S0), I(S1), I(S2)), x.device, 'unet_train')
        y = torch.empty_like(x)
        e0 = _prof_begin()
        kept = ctypes.c_uint(0)  # which three-term input copies the forward left in `saved`: travels with this context
        check(L.nc_unet_deconv_train_fwd(_ptr(packed), _ptr(x), _ptr(y), _ptr(saved), I(N), I(S0), I(S1), I(S2), _ptr(ws),
                                         Z(ws.numel()), _stream(), ctypes.byref(kept)), 'nc_unet_deconv_train_fwd')
        ctx.kept = kept.value
        if e0 is not None:
            _prof_end(e0, 'unet_fwd', 2.0 * 663809 * x.numel())
        ctx.save_for_backward(x, y, saved)
        ctx.packed = packed
        ctx.packed_gen = _param_generation(packed)
        ctx.shapes = [tuple(p.shape) for p in params]
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, saved = ctx.saved_tensors
        if _param_generation(ctx.packed) != ctx.packed_gen:
            raise _lib.NcError('fused Unet_deconv: the parameters were updated between this forward and its backward')
        dy = dy.contiguous()
        N, _, S0, S1, S2 = x.shape
        L = lib()
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        dpar = _grad_target(ctx, 1)
        ws = workspace(L.nc_unet_deconv_train_ws_bytes(I(N), I(S0), I(S1), I(S2)), x.device, 'unet_train')
        e0 = _prof_begin()
        check(L.nc_unet_deconv_bwd(_ptr(ctx.packed), _ptr(x), _ptr(y), _ptr(saved), _ptr(dy), _ptr(dx), _ptr(dpar), I(N),
                                   I(S0), I(S1), I(S2), _ptr(ws), Z(ws.numel()), _stream(), ctypes.c_uint(ctx.kept)), 'nc_unet_deconv_bwd')
        if e0 is not None:  # dgrad + wgrad of every layer but the first one's data gradient
            _prof_end(e0, 'unet_bwd', 2.0 * (2 * 663809 - 1728) * x.numel())
        return (dx,) + tuple(_param_grads(ctx, dpar, ctx.shapes, 1))


def unet_deconv_train(x, params):
    """Unet_deconv.forward (networks.py:512-538) with autograd, as one C call per direction."""
    return _UnetDeconvTrain.apply(x, *params)


class _DeepLinear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, *params):
        x = x.contiguous()
        _chk(x, *params)
        _f32(x, *params)
        N, _, S0, S1, S2 = x.shape
        L = lib()
        packed = _pack_params(params)
        if packed.numel() != L.nc_deep_linear_param_floats():
            raise _lib.NcError('fused DeepLinearGenerator: parameter count does not match')
        need = any(ctx.needs_input_grad)
        saved = torch.empty(L.nc_deep_linear_saved_floats(I(N), I(S0), I(S1), I(S2)), dtype=torch.float32,
                            device=x.device) if need else None
        ws = workspace(L.nc_deep_linear_ws_bytes(I(N), I(S0), I(S1), I(S2)), x.device, 'deep_linear')
        y = torch.empty_like(x)
        e0 = _prof_begin()
        kept = ctypes.c_uint(0)
        check(L.nc_deep_linear_fwd(_ptr(packed), _ptr(x), _ptr(y), _ptr(saved), I(N), I(S0), I(S1), I(S2), _ptr(ws),
                                   Z(ws.numel()), _stream(), ctypes.byref(kept)), 'nc_deep_linear_fwd')
        ctx.kept = kept.value
        if e0 is not None:
            _prof_end(e0, 'deep_linear_fwd', 2.0 * 647120 * x.numel())
        if need:
            ctx.save_for_backward(x, saved)
            ctx.packed = packed
            ctx.packed_gen = _param_generation(packed)
            ctx.shapes = [tuple(p.shape) for p in params]
        return y

    @staticmethod
    def backward(ctx, dy):
        x, saved = ctx.saved_tensors
        if _param_generation(ctx.packed) != ctx.packed_gen:
            raise _lib.NcError('fused DeepLinearGenerator: the parameters were updated between this forward and its backward')
        dy = dy.contiguous()
        N, _, S0, S1, S2 = x.shape
        L = lib()
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        dpar = _grad_target(ctx, 1)
        ws = workspace(L.nc_deep_linear_ws_bytes(I(N), I(S0), I(S1), I(S2)), x.device, 'deep_linear')
        e0 = _prof_begin()
        check(L.nc_deep_linear_bwd(_ptr(ctx.packed), _ptr(x), _ptr(saved), _ptr(dy), _ptr(dx), _ptr(dpar), I(N), I(S0), I(S1),
                                   I(S2), _ptr(ws), Z(ws.numel()), _stream(), ctypes.c_uint(ctx.kept)), 'nc_deep_linear_bwd')
        if e0 is not None:
            _prof_end(e0, 'deep_linear_bwd', 2.0 * (2 * 647120 - (0 if dx is not None else 21952)) * x.numel())
        return (dx,) + tuple(_param_grads(ctx, dpar, ctx.shapes, 1))


def deep_linear(x, params):
    """DeepLinearGenerator.forward (networks.py:913-917), with autograd, as one C call per direction."""
    return _DeepLinear.apply(x, *params)


# ---- the same two generators on the 16-bit end-to-end path (nc_unet_deconv_lp_* / nc_deep_linear_lp_*) ---------------
def gen_lp_supported(kind, shape):
    """True when --precision bf16 can run this generator as the whole-network 16-bit call (every layer covered)."""
    if conv_precision != 'bf16' or len(shape) != 5 or shape[1] != 1:
        return False
    N, _, S0, S1, S2 = shape
    fn = lib().nc_unet_deconv_lp_supported if kind == 'unet' else lib().nc_deep_linear_lp_supported
    return bool(fn(I(N), I(S0), I(S1), I(S2), I(_DT['bf16'])))


class _GenLp(torch.autograd.Function):
    """kind 'unet': Unet_deconv, 'linear': DeepLinearGenerator; bf16 activations end to end, fp32 master weights."""

    @staticmethod
    def forward(ctx, x, kind, *params):
        x = x.contiguous()
        _chk(x, *params)
        _f32(x, *params)
        N, _, S0, S1, S2 = x.shape
        L = lib()
        pre = 'nc_unet_deconv_lp' if kind == 'unet' else 'nc_deep_linear_lp'
        packed = _pack_params(params)
        want = L.nc_unet_deconv_param_floats() if kind == 'unet' else L.nc_deep_linear_param_floats()
        if packed.numel() != want:
            raise _lib.NcError('%s: parameter count does not match' % pre)
        dims = (I(N), I(S0), I(S1), I(S2))
        nsv = getattr(L, pre + '_saved_bytes')(*dims)
        if nsv == 0:
            raise _lib.NcError('%s: shape %s is not covered by the 16-bit kernels' % (pre, (S0, S1, S2)))
        saved = torch.empty(nsv, dtype=torch.uint8, device=x.device)
        ws = workspace(getattr(L, pre + '_ws_bytes')(*dims), x.device, pre)
        y = torch.empty_like(x)
        dt = _DT['bf16']
        flop = 2.0 * (663809 if kind == 'unet' else 647120) * x.numel()
        e0 = _prof_begin()
        kept = ctypes.c_uint(0)  # deep_linear_gen: the form the forward took (the backward follows it, not the switches of its own moment)
        extra = (ctypes.byref(kept),) if kind != 'unet' else ()
        check(getattr(L, pre + '_fwd')(_ptr(packed), _ptr(x), _ptr(y), _ptr(saved), *dims, I(dt), _ptr(ws), Z(ws.numel()),
                                       _stream(), *extra), pre + '_fwd')
        ctx.kept = kept.value
        if e0 is not None:
            _prof_end(e0, ('unet' if kind == 'unet' else 'deep_linear') + '_lp_fwd', flop)
        ctx.save_for_backward(x, y, saved)
        ctx.packed, ctx.packed_gen, ctx.kind = packed, _param_generation(packed), kind
        ctx.shapes = [tuple(p.shape) for p in params]
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, saved = ctx.saved_tensors
        if _param_generation(ctx.packed) != ctx.packed_gen:
            raise _lib.NcError('16-bit generator: the parameters were updated between this forward and its backward')
        dy = dy.contiguous()
        N, _, S0, S1, S2 = x.shape
        L = lib()
        kind = ctx.kind
        pre = 'nc_unet_deconv_lp' if kind == 'unet' else 'nc_deep_linear_lp'
        dims = (I(N), I(S0), I(S1), I(S2))
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        dpar = _grad_target(ctx, 2)
        ws = workspace(getattr(L, pre + '_ws_bytes')(*dims), x.device, pre)
        dt = _DT['bf16']
        e0 = _prof_begin()
        if kind == 'unet':
            check(L.nc_unet_deconv_lp_bwd(_ptr(ctx.packed), _ptr(x), _ptr(y), _ptr(saved), _ptr(dy), _ptr(dx), _ptr(dpar), *dims,
                                          I(dt), _ptr(ws), Z(ws.numel()), _stream()), pre + '_bwd')
        else:
            check(L.nc_deep_linear_lp_bwd(_ptr(ctx.packed), _ptr(x), _ptr(saved), _ptr(dy), _ptr(dx), _ptr(dpar), *dims, I(dt),
                                          _ptr(ws), Z(ws.numel()), _stream(), ctypes.c_uint(ctx.kept)), pre + '_bwd')
        if e0 is not None:
            mac = (2 * 663809 - 1728) if kind == 'unet' else 2 * 647120
            _prof_end(e0, ('unet' if kind == 'unet' else 'deep_linear') + '_lp_bwd', 2.0 * mac * x.numel())
        return (dx, None) + tuple(_param_grads(ctx, dpar, ctx.shapes, 2))


def unet_deconv_lp(x, params):
    return _GenLp.apply(x, 'unet', *params)


def deep_linear_lp(x, params):
    return _GenLp.apply(x, 'linear', *params)


# ---- torch.nn.utils.spectral_norm (NLayerDiscriminatorSN, networks.py:1069-1111) ------------------------------------
class _SpectralNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, w_orig, u, v, power_iteration):
        w_orig = w_orig.contiguous()
        _chk(w_orig, u, v)
        _f32(w_orig, u, v)
        K = w_orig.shape[0]
        M = w_orig.numel() // K
        if u.numel() != K or v.numel() != M:
            raise _lib.NcError('spectral_norm: u / v do not match the weight (%d x %d)' % (K, M))
        w = torch.empty_like(w_orig)
        sigma = torch.empty(1, dtype=torch.float32, device=w_orig.device)
        scratch = torch.empty(K, dtype=torch.float32, device=w_orig.device)
        check(lib().nc_spectral_norm_fwd(_ptr(w_orig), _ptr(u), _ptr(v), _ptr(w), _ptr(sigma), _ptr(scratch), I(K), I(M),
                                         I(1 if power_iteration else 0), F(1e-12), _stream()), 'nc_spectral_norm_fwd')
        # u, v as they stand after this forward (later forwards update the buffers in place)
        ctx.save_for_backward(w, u.clone(), v.clone(), sigma)
        return w

    @staticmethod
    def backward(ctx, g):
        w, u, v, sigma = ctx.saved_tensors
        g = g.contiguous()
        K = w.shape[0]
        M = w.numel() // K
        dw = torch.empty_like(w)
        check(lib().nc_spectral_norm_bwd(_ptr(g), _ptr(w), _ptr(u), _ptr(v), _ptr(sigma), _ptr(dw), I(K), I(M), _stream()),
              'nc_spectral_norm_bwd')
        return dw, None, None, None


def spectral_norm_weight(w_orig, u, v, power_iteration):
    """weight = w_orig / sigma with one power iteration on (u, v) (in place) when `power_iteration`."""
    return _SpectralNorm.apply(w_orig, u, v, bool(power_iteration))
