"""NeRF MLP (mirror of nerf-ours/model.py:8-63) whose parameters are views into ONE flat
fp32 buffer laid out in `model.parameters()` order, so that

  * the HIP kernels (fastnerf_mlp_fwd / _bwd / adam_step) work on the flat buffer directly,
  * `state_dict()` keeps the reference's names (pts_linears.{0..7}, views_linears.0,
    feature_linear, alpha_linear, rgb_linear) and `load_state_dict` of a reference
    checkpoint (with or without the DataParallel `module.` prefix) just works,
  * a data-parallel all-reduce is a single collective over `flat_grad`.

Only the configuration the reference's configs use is implemented natively
(D=8, W=256, skips=[4], use_viewdirs=True, input_ch=63, input_ch_views=27).
"""
import torch
from torch import nn

from . import ops

SHAPES = (
    [(f'pts_linears.{i}', (256, 63 if i == 0 else (319 if i == 5 else 256))) for i in range(8)]
    + [('views_linears.0', (128, 283)), ('feature_linear', (256, 256)), ('alpha_linear', (1, 256)),
       ('rgb_linear', (3, 128))]
)


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
        if not (D == 8 and W == 256 and input_ch == 63 and input_ch_views == 27 and list(skips) == [4]
                and use_viewdirs):
            raise NotImplementedError('the HIP MLP implements D=8, W=256, input_ch=63, input_ch_views=27, '
                                      'skips=[4], use_viewdirs=True (the configuration of every nerf-ours config)')
        self.D, self.W, self.input_ch, self.input_ch_views = D, W, input_ch, input_ch_views
        self.skips, self.use_viewdirs = list(skips), use_viewdirs
        # same construction order as the reference => identical init under torch.manual_seed
        pts = nn.ModuleList([nn.Linear(input_ch, W)] +
                            [nn.Linear(W, W) if i not in self.skips else nn.Linear(W + input_ch, W)
                             for i in range(D - 1)])
        views = nn.ModuleList([nn.Linear(input_ch_views + W, W // 2)])
        feature, alpha, rgb = nn.Linear(W, W), nn.Linear(W, 1), nn.Linear(W // 2, 3)
        self.pts_linears, self.views_linears = pts, views
        self.feature_linear, self.alpha_linear, self.rgb_linear = feature, alpha, rgb
        dev = torch.device(device)
        self.flat = flat if flat is not None else torch.empty(ops.NET_PARAMS, device=dev, dtype=torch.float32)
        self.flat_grad = flat_grad if flat_grad is not None else torch.zeros(ops.NET_PARAMS, device=dev,
                                                                             dtype=torch.float32)
        mods = dict(self.named_modules())
        for name, off, shape in param_slices():
            mod_name, leaf = name.rsplit('.', 1)
            mod = mods[mod_name]
            n = 1
            for s in shape:
                n *= s
            view = self.flat[off:off + n].view(shape)
            with torch.no_grad():
                view.copy_(getattr(mod, leaf).detach().to(dev))
            p = nn.Parameter(view)
            p.grad = self.flat_grad[off:off + n].view(shape)
            setattr(mod, leaf, p)
        self._packed = None

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
        alpha = self.alpha_linear(h)
        h = torch.cat([self.feature_linear(h), views], -1)
        h = F.relu(self.views_linears[0](h))
        return torch.cat([self.rgb_linear(h), alpha], -1)

    def load_state_dict(self, state_dict, strict=True):
        sd = {(k[7:] if k.startswith('module.') else k): v for k, v in state_dict.items()}
        return super().load_state_dict(sd, strict=strict)
