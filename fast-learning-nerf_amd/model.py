"""NeRF MLP (mirror of nerf-ours/model.py:8-63) whose parameters are views into ONE flat
fp32 buffer laid out in `model.parameters()` order, so that

  * the HIP kernels (fastnerf_mlp_fwd / _bwd / adam_step) work on the flat buffer directly,
  * `state_dict()` keeps the reference's names (pts_linears.{0..7}, views_linears.0,
    feature_linear, alpha_linear, rgb_linear) and `load_state_dict` of a reference
    checkpoint (with or without the DataParallel `module.` prefix) just works,
  * a data-parallel all-reduce is a single collective over `flat_grad`.

The kernels implement the configuration the reference's configs use (D=8, W=256, skips=[4], use_viewdirs=True,
input_ch=63, input_ch_views=27).  `use_viewdirs=False` (model.py:35-36,60-61: one `output_linear` 256 -> output_ch on the
trunk, no view branch) runs on the SAME kernels through an exactly equivalent view-branch network (`NeRF._sync_kernel_net`):
feature layer = identity, view layer rows (2c, 2c+1) = (+w_c, -w_c) of output row c with zero weights on the direction
encoding, colour head = relu(a) - relu(-a) = a, sigma head = output row 3.  In fp32 arithmetic every one of those steps is
exact (products with 0 / 1, one non-zero term per sum), so the result is the reference's dot product; the parameters,
their names, order and gradients are the reference's (`views_linears.0` exists there too, unused).
"""
import torch
from torch import nn

from . import ops

SHAPES = (
    [(f'pts_linears.{i}', (256, 63 if i == 0 else (319 if i == 5 else 256))) for i in range(8)]
    + [('views_linears.0', (128, 283)), ('feature_linear', (256, 256)), ('alpha_linear', (1, 256)),
       ('rgb_linear', (3, 128))]
)


def noview_shapes(output_ch):
    """Modules of the reference model without view directions in registration order (model.py:20-36, input_ch_views = 0)."""
    return ([(f'pts_linears.{i}', (256, 63 if i == 0 else (319 if i == 5 else 256))) for i in range(8)]
            + [('views_linears.0', (128, 256)), ('output_linear', (output_ch, 256))])


def _slices(shapes):
    out, off = [], 0
    for name, (o, i) in shapes:
        out.append((name + '.weight', off, (o, i)))
        off += o * i
        out.append((name + '.bias', off, (o,)))
        off += o
    return out, off


def noview_slices(output_ch):
    """[(name, offset, shape)], total floats -- parameters() order of the model without view directions."""
    return _slices(noview_shapes(output_ch))


PTS_FLOATS = sum(o * i + o for _, (o, i) in SHAPES[:8])   # the trunk comes first in both layouts


def param_slices():
    """[(name, offset, shape)] in model.parameters() order."""
    out, off = [], 0
    for name, (o, i) in SHAPES:
        out.append((name + '.weight', off, (o, i)))
        off += o * i
        out.append((name + '.bias', off, (o,)))
        off += o
    assert off == ops.NET_PARAMS
    return out


class NeRF(nn.Module):
    def __init__(self, D=8, W=256, input_ch=63, input_ch_views=27, output_ch=4, skips=[4], use_viewdirs=True,
                 device='cuda', flat=None, flat_grad=None):
        super().__init__()
        if not (D == 8 and W == 256 and input_ch == 63 and list(skips) == [4]
                and ((use_viewdirs and input_ch_views == 27) or (not use_viewdirs and input_ch_views == 0))):
            raise NotImplementedError('the HIP MLP implements D=8, W=256, input_ch=63, skips=[4] with input_ch_views=27 '
                                      '(use_viewdirs) or 0 (no view directions)')
        self.D, self.W, self.input_ch, self.input_ch_views = D, W, input_ch, input_ch_views
        self.skips, self.use_viewdirs, self.output_ch = list(skips), use_viewdirs, output_ch
        # same construction order as the reference => identical init under torch.manual_seed
        pts = nn.ModuleList([nn.Linear(input_ch, W)] +
                            [nn.Linear(W, W) if i not in self.skips else nn.Linear(W + input_ch, W)
                             for i in range(D - 1)])
        views = nn.ModuleList([nn.Linear(input_ch_views + W, W // 2)])
        self.pts_linears, self.views_linears = pts, views
        if use_viewdirs:
            feature, alpha, rgb = nn.Linear(W, W), nn.Linear(W, 1), nn.Linear(W // 2, 3)
            self.feature_linear, self.alpha_linear, self.rgb_linear = feature, alpha, rgb
            slices, total = param_slices(), ops.NET_PARAMS
        else:
            if output_ch < 4:
                raise ValueError('output_ch must be >= 4 (rgb + sigma)')
            self.output_linear = nn.Linear(W, output_ch)
            slices, total = noview_slices(output_ch)
        dev = torch.device(device)
        pflat = flat if flat is not None else torch.empty(total, device=dev, dtype=torch.float32)
        pgrad = flat_grad if flat_grad is not None else torch.zeros(total, device=dev, dtype=torch.float32)
        assert pflat.numel() == total and pgrad.numel() == total
        mods = dict(self.named_modules())
        for name, off, shape in slices:
            mod_name, leaf = name.rsplit('.', 1)
            mod = mods[mod_name]
            n = 1
            for s in shape:
                n *= s
            view = pflat[off:off + n].view(shape)
            with torch.no_grad():
                view.copy_(getattr(mod, leaf).detach().to(dev))
            p = nn.Parameter(view)
            p.grad = pgrad[off:off + n].view(shape)
            setattr(mod, leaf, p)
        self.param_flat, self.param_grad = pflat, pgrad      # the reference's parameters, parameters() order
        if use_viewdirs:
            self.flat, self.flat_grad = pflat, pgrad          # ... which is also what the kernels read / write
        else:
            # what the kernels read / write: the equivalent view-branch network (module docstring), standard layout
            self.flat = torch.zeros(ops.NET_PARAMS, device=dev, dtype=torch.float32)
            self.flat_grad = torch.zeros(ops.NET_PARAMS, device=dev, dtype=torch.float32)
            k = self._kernel_views(self.flat)
            k['feature_linear.weight'].copy_(torch.eye(256, device=dev))
            for c in range(3):
                k['rgb_linear.weight'][c, 2 * c] = 1.0
                k['rgb_linear.weight'][c, 2 * c + 1] = -1.0
            self._sync_kernel_net()
        self._packed = None

    # ---- no view directions: parameters <-> the equivalent view-branch network the kernels run -----------------
    @staticmethod
    def _kernel_views(flat):
        out = {}
        for name, off, shape in param_slices():
            n = 1
            for s in shape:
                n *= s
            out[name] = flat[off:off + n].view(shape)
        return out

    def _sync_kernel_net(self):
        if self.use_viewdirs:
            return
        with torch.no_grad():
            k = self._kernel_views(self.flat)
            self.flat[:PTS_FLOATS].copy_(self.param_flat[:PTS_FLOATS])
            w, b = self.output_linear.weight, self.output_linear.bias
            vw, vb = k['views_linears.0.weight'], k['views_linears.0.bias']
            vw[0:6:2, :256] = w[:3]
            vw[1:6:2, :256] = -w[:3]
            vb[0:6:2] = b[:3]
            vb[1:6:2] = -b[:3]
            k['alpha_linear.weight'].copy_(w[3:4])
            k['alpha_linear.bias'].copy_(b[3:4])

    def param_grads_from(self, kernel_grad):
        """Gradients in parameters() order from a gradient buffer the kernels wrote (standard layout)."""
        k = self._kernel_views(kernel_grad)
        if self.use_viewdirs:
            return [k[name] for name, _, _ in param_slices()]
        out = []
        for name, _, shape in noview_slices(self.output_ch)[0]:
            if name.startswith('pts_linears.'):
                out.append(k[name])
            elif name.startswith('views_linears.'):          # never used by the forward pass (model.py:60-61)
                out.append(torch.zeros(shape, device=kernel_grad.device))
            else:
                vg = k['views_linears.0.weight'] if name.endswith('weight') else k['views_linears.0.bias']
                ag = k['alpha_linear.weight'] if name.endswith('weight') else k['alpha_linear.bias']
                g = torch.zeros(shape, device=kernel_grad.device)
                if name.endswith('weight'):
                    g[:3] = vg[0:6:2, :256] - vg[1:6:2, :256]
                else:
                    g[:3] = vg[0:6:2] - vg[1:6:2]
                g[3:4] = ag                                   # rows >= 4 (output_ch = 5) are never read: zero gradient
                out.append(g)
        return out

    def collect_grads(self):
        """After a backward that wrote `flat_grad`: make the parameters' .grad (views of `param_grad`) current."""
        if self.use_viewdirs:
            return
        with torch.no_grad():
            for (name, off, shape), g in zip(noview_slices(self.output_ch)[0], self.param_grads_from(self.flat_grad)):
                self.param_grad[off:off + g.numel()].view(shape).copy_(g)

    # ---- packed weights for the MFMA kernels -----------------------------------------------
    def packed(self, refresh=True):
        """(packed_fwd, packed_bwd) fragment-ordered copies of the weights.  Re-packed from the
        flat buffer on every call unless refresh=False (one ~5 MB launch; callers that update
        the weights themselves, e.g. the fused Trainer, pass refresh=False between updates)."""
        if self._packed is None or self._packed_mode != ops.get_math():
            self._packed = (torch.empty(ops.packed_floats(0, 1), device=self.flat.device),
                            torch.empty(ops.packed_floats(0, 2), device=self.flat.device))
            self._packed_mode = ops.get_math()
            refresh = True
        if refresh:
            self._sync_kernel_net()
            ops.mlp_pack(self.flat, *self._packed)
        return self._packed

    def forward(self, x):
        """Reference signature: x = [.., 63 + 27] already-embedded inputs (model.py:38-63).
        Convenience path for third-party callers (e.g. mesh extraction); the renderer never
        calls it -- it runs the fused HIP kernels on raw rays instead."""
        F = torch.nn.functional
        pts, views = torch.split(x, [self.input_ch, self.input_ch_views], dim=-1)
        h = pts
        for i in range(self.D):
            h = F.relu(self.pts_linears[i](h))
            if i in self.skips:
                h = torch.cat([pts, h], -1)
        if not self.use_viewdirs:
            return self.output_linear(h)
        alpha = self.alpha_linear(h)
        h = torch.cat([self.feature_linear(h), views], -1)
        h = F.relu(self.views_linears[0](h))
        return torch.cat([self.rgb_linear(h), alpha], -1)

    def load_state_dict(self, state_dict, strict=True):
        sd = {(k[7:] if k.startswith('module.') else k): v for k, v in state_dict.items()}
        return super().load_state_dict(sd, strict=strict)
