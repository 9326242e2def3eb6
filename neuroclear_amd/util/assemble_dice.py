"""Assemble_Dice on the device (reference: util/assemble_dice.py:11-213).

Same interface -- Assemble_Dice(opt), addToStack(visuals), assemble_all(), getDict() -- but the cubes never leave HBM:
addToStack crops the border and overlap-adds cube/8 into a padded fp32 accumulator right away (nc_assemble_scatter_add,
in arrival = index order, exactly the reference's summation order), assemble_all runs nc_assemble_finalize
((acc / count) * 8 * 65535, truncating cast, crop of the dicing pad) with the count computed analytically.
`--normalize_intensity` (:188-192) runs on the device too: exact percentiles of the merged, still padded volume by radix
select (util/percentile.py restates np.percentile of the reference's numpy 1.21.2), then the arithmetic of
skimage.exposure.rescale_intensity (0.18.3) on a float32 image -- third-party arithmetic restated from its source,
parity unpinned (scikit-image is not installed here).  `--histogram_match` (:149-151, skimage match_histograms of every
border-cropped output cube against its input cube) is nc_match_histograms: radix sort of both cubes + np.interp's
arithmetic in float64 on the device, restated from scikit-image 0.18.3, parity unpinned for the same reason."""
from collections import OrderedDict

import numpy as np
import torch

from .._lib import I, P, check, lib
from . import util


def match_histograms(source, template):
    """skimage.exposure.match_histograms(source, template) on the device (float32 tensors of equal size)."""
    from .._lib import L_, Z
    src = source.contiguous().float()
    tm = template.contiguous().float()
    if src.numel() != tm.numel() or not src.is_cuda or not tm.is_cuda:
        raise ValueError('match_histograms: two CUDA tensors of equal size expected')
    n = src.numel()
    out = torch.empty_like(src)
    nb = lib().nc_match_histograms_ws_bytes(L_(n))
    ws = torch.empty(nb, dtype=torch.uint8, device=src.device)
    check(lib().nc_match_histograms(P(src.data_ptr()), P(tm.data_ptr()), P(out.data_ptr()), L_(n), P(ws.data_ptr()), Z(nb),
                                    P(torch.cuda.current_stream().cuda_stream)), 'nc_match_histograms')
    return out


def match_cube(fake, real, roi_size, border_cut, device):
    """util/assemble_dice.py:149-151: the border-cropped fake cube matched to the border-cropped real cube; returns the
    (R+2b)^3 fake cube with its interior replaced (the border is dropped by add_cube anyway)."""
    E, b = roi_size + 2 * border_cut, border_cut
    fake = fake.reshape(E, E, E).to(device, torch.float32)
    real = real.reshape(E, E, E).to(device, torch.float32)
    out = fake.clone()
    out[b:-b, b:-b, b:-b] = match_histograms(fake[b:-b, b:-b, b:-b], real[b:-b, b:-b, b:-b])
    return out


class Assemble_Dice:
    def __init__(self, opt, image_size_original=None, slab=None):
        """image_size_original: (z, y, x) of the un-padded volume; when omitted it is taken from opt.volume_shape.
        slab = (za, zb): the accumulators hold only the padded planes [za, zb) (sharded inference, test_dice.py assemble='slab':
        a rank's cubes touch two or three z-layers); add_cube then accepts only cubes inside that range."""
        if image_size_original is None:
            image_size_original = opt.volume_shape
        self.image_size_original = tuple(int(s) for s in image_size_original)
        self.border_cut = opt.border_cut
        self.roi_size = opt.dice_size[0]
        self.overlap = opt.overlap
        if self.border_cut < 1:
            raise ValueError('border_cut must be >= 1 (util/assemble_dice.py:143-145 crops cube[b:-b])')
        if self.overlap < 1:
            raise ValueError('overlap must be >= 1: the reference assembler adds nothing for overlap 0 '
                             '(util/assemble_dice.py:170) and returns an all-zero volume')
        self.histogram_match = bool(getattr(opt, 'histogram_match', False))
        self.normalize_intensity = bool(getattr(opt, 'normalize_intensity', False))
        self.p1, self.p99 = getattr(opt, 'sat_level', [0.25, 99.75])  # options/test_options.py:25
        self.step = self.roi_size - self.overlap
        self.image_size = util.padded_shape(self.image_size_original, self.roi_size, self.overlap)
        self.z_steps, self.y_steps, self.x_steps = util.grid_steps(self.image_size, self.roi_size, self.overlap)
        self.len_cube_queue = self.z_steps * self.y_steps * self.x_steps
        self.visual_names = ['real', 'fake']
        self.imtype = opt.data_type
        if self.imtype not in ('uint8', 'uint16'):
            raise ValueError('data_type must be uint8 or uint16')
        self.skip_real = opt.skip_real
        self.device = torch.device('cuda', opt.gpu_ids[0]) if getattr(opt, 'gpu_ids', None) else torch.device('cuda')
        self.acc = OrderedDict()
        self.count = OrderedDict()
        self.visual_ret = OrderedDict()
        self.slab = None if slab is None else (int(slab[0]), int(slab[1]))
        shape = self.image_size if self.slab is None else (self.slab[1] - self.slab[0],) + tuple(self.image_size[1:])
        for name in self.visual_names:
            if self.skip_real and name == 'real':
                continue
            self.acc[name] = torch.zeros(shape, dtype=torch.float32, device=self.device)
            self.count[name] = 0

    def indexTo3DIndex(self, index):
        return (index // (self.x_steps * self.y_steps), (index % (self.x_steps * self.y_steps)) // self.x_steps,
                index % self.x_steps)

    def indexToCoordinates(self, index):
        z, y, x = self.indexTo3DIndex(index)
        return z * self.step, y * self.step, x * self.step

    def add_cube(self, name, cube, index):
        """Overlap-add one (R+2b)^3 network output at cube position `index`."""
        E = self.roi_size + 2 * self.border_cut
        cube = cube.reshape(-1)
        if cube.numel() != E * E * E:
            raise AssertionError('the cube dimensions are invalid.')
        if cube.dtype != torch.float32 or not cube.is_cuda:
            cube = cube.to(self.device, torch.float32)
        cube = cube.contiguous()
        P0, P1, P2 = self.image_size
        base = self.acc[name].data_ptr()
        if self.slab is not None:  # the kernel indexes absolute planes: hand it where plane 0 WOULD be (it only touches the cube's planes)
            z0 = self.indexToCoordinates(int(index))[0]
            if z0 < self.slab[0] or z0 + self.roi_size > self.slab[1]:
                raise AssertionError('cube %d lies outside the planes [%d, %d) of this accumulator' % (index, self.slab[0], self.slab[1]))
            base -= self.slab[0] * P1 * P2 * 4
        check(lib().nc_assemble_scatter_add(P(cube.data_ptr()), P(base), I(P0), I(P1), I(P2),
                                            I(self.roi_size), I(self.overlap), I(self.border_cut), I(int(index)),
                                            P(torch.cuda.current_stream().cuda_stream)), 'nc_assemble_scatter_add')

    def match_cube(self, fake, real):
        return match_cube(fake, real, self.roi_size, self.border_cut, self.device)

    def addToStack(self, cube):
        """cube: OrderedDict {'real': [1,1,E,E,E], 'fake': ...} as returned by model.get_current_visuals()."""
        if self.histogram_match:
            cube = OrderedDict(cube)
            cube['fake'] = self.match_cube(cube['fake'], cube['real'])
        for name in self.visual_names:
            if self.skip_real and name == 'real':
                continue
            self.add_cube(name, cube[name], self.count[name])
            self.count[name] += 1

    def finalize_slab(self, own_acc, za, zb):
        """(acc / count) * 8 * scale and the truncating cast for the padded planes [za, zb) held in own_acc (already complete: every
        rank's contributions added): returns the device tensor of the planes [za, zb) & [0, L0) of the un-padded volume."""
        L0, L1, L2 = self.image_size_original
        P0, P1, P2 = self.image_size
        z0, z1 = min(za, L0), min(zb, L0)
        u16 = self.imtype == 'uint16'
        out = torch.empty((z1 - z0, L1, L2), dtype=torch.int16 if u16 else torch.uint8, device=self.device)
        if z1 > z0:
            check(lib().nc_assemble_finalize_slab(P(own_acc.data_ptr()), P(out.data_ptr()), I(1 if u16 else 0), I(P0), I(P1), I(P2), I(L0),
                                                  I(L1), I(L2), I(self.roi_size), I(self.overlap), I(z0), I(z1 - z0), I(za),
                                                  P(torch.cuda.current_stream().cuda_stream)), 'nc_assemble_finalize_slab')
        return out

    def assemble_all(self):
        if self.slab is not None:
            raise Exception('a slab accumulator is finalised by its owner ranks (finalize_slab), not by assemble_all')
        L0, L1, L2 = self.image_size_original
        P0, P1, P2 = self.image_size
        for name, acc in self.acc.items():
            if self.count[name] != self.len_cube_queue:
                raise Exception('expected %d cubes for %s, got %d' % (self.len_cube_queue, name, self.count[name]))
            u16 = self.imtype == 'uint16'
            out = torch.empty((L0, L1, L2), dtype=torch.int16 if u16 else torch.uint8, device=self.device)
            if self.normalize_intensity:
                self._finalize_normalized(acc, out, u16)
                host = out.cpu().numpy()
                self.visual_ret[name] = host.view(np.uint16) if u16 else host
                continue
            check(lib().nc_assemble_finalize(P(acc.data_ptr()), P(out.data_ptr()), I(1 if u16 else 0), I(P0), I(P1),
                                             I(P2), I(L0), I(L1), I(L2), I(self.roi_size), I(self.overlap),
                                             P(torch.cuda.current_stream().cuda_stream)), 'nc_assemble_finalize')
            host = out.cpu().numpy()
            self.visual_ret[name] = host.view(np.uint16) if u16 else host

    def _finalize_normalized(self, acc, out, u16):
        """util/assemble_dice.py:184-203 with --normalize_intensity: percentiles over the merged PADDED volume."""
        from . import percentile as pct
        L0, L1, L2 = self.image_size_original
        stream = P(torch.cuda.current_stream().cuda_stream)
        merged = torch.empty_like(acc)
        check(lib().nc_assemble_merge(P(acc.data_ptr()), P(merged.data_ptr()), I(L0), I(L1), I(L2), I(self.roi_size),
                                      I(self.overlap), stream), 'nc_assemble_merge')
        p_lo, p_hi = pct.percentile(merged, (self.p1, self.p99))
        if not p_hi > p_lo:
            raise ValueError('normalize_intensity: empty intensity range (%g, %g)' % (p_lo, p_hi))
        self.last_percentiles = (p_lo, p_hi)
        from .._lib import F
        check(lib().nc_assemble_rescale_finalize(P(merged.data_ptr()), P(out.data_ptr()), I(1 if u16 else 0), I(L0),
                                                 I(L1), I(L2), I(self.roi_size), I(self.overlap),
                                                 F(np.float32(p_lo)), F(np.float32(p_hi)), F(np.float32(p_hi - p_lo)),
                                                 stream), 'nc_assemble_rescale_finalize')

    def getDict(self):
        return self.visual_ret
