"""Mirror of nerf-ours/run_nerf_helpers.py on the HIP ops (same names, arguments, returns).

get_rays :68-78, get_rays_np :81-88, ndc_rays :91-108, get_embedder :48-63, sample_pdf :112-155,
img2mse/mse2psnr/to8b :9-11, compute_ssim :158-234 (evaluation metric, plain torch on whatever device the images
live on).  The ray / sampling functions need GPU tensors; there is no CPU fallback for them."""
import numpy as np
import torch

from . import ops

img2mse = lambda x, y: torch.mean((x - y) ** 2)
mse2psnr = lambda x: -10. * torch.log(x) / torch.log(torch.tensor([10.], device=x.device if torch.is_tensor(x) else None))
to8b = lambda x: (255 * np.clip(x, 0, 1)).astype(np.uint8)


class Embedder:
    """Sinusoidal positional encoding (run_nerf_helpers.py:15-45).  Only the
    configuration every reference call site uses is implemented natively:
    include_input, log_sampling, periodic_fns=[sin, cos], input_dims=3."""

    def __init__(self, **kwargs):
        self.kwargs = kwargs
        if not (kwargs.get('include_input', True) and kwargs.get('log_sampling', True)
                and kwargs.get('input_dims', 3) == 3):
            raise NotImplementedError('HIP embedder implements include_input + log_sampling + 3-D inputs')
        self.num_freqs = int(kwargs['num_freqs'])
        if int(kwargs.get('max_freq_log2', self.num_freqs - 1)) != self.num_freqs - 1:
            raise NotImplementedError('max_freq_log2 must equal num_freqs-1 (bands are exactly 2^k)')
        self.out_dim = 3 + 6 * self.num_freqs

    def embed(self, inputs):
        return ops.posenc(inputs, self.num_freqs)


def get_embedder(multires, i=0):
    if i == -1:
        return torch.nn.Identity(), 3
    eo = Embedder(include_input=True, input_dims=3, max_freq_log2=multires - 1, num_freqs=multires,
                  log_sampling=True, periodic_fns=[torch.sin, torch.cos])
    embed = lambda x, eo=eo: eo.embed(x)
    embed.num_freqs = multires
    return embed, eo.out_dim


def get_rays(H, W, K, c2w):
    return ops.gen_rays(H, W, K, c2w)


def get_rays_np(H, W, K, c2w):
    ro, rd = ops.gen_rays(H, W, K, torch.as_tensor(np.asarray(c2w), dtype=torch.float32))
    return ro.cpu().numpy(), rd.cpu().numpy()


def ndc_rays(H, W, focal, near, rays_o, rays_d):
    return ops.ndc_rays(H, W, focal, near, rays_o, rays_d)


def sample_pdf(bins, weights, N_samples, det=False, pytest=False):
    """Inverse-CDF sampling (run_nerf_helpers.py:112-155) on the device.  `pytest=True` reproduces
    the reference's hook: u = np.random.seed(0); np.random.rand(...) (or linspace when det)."""
    lead = list(bins.shape[:-1])
    u = None
    if pytest:
        np.random.seed(0)
        if det:
            u = np.broadcast_to(np.linspace(0., 1., N_samples), lead + [N_samples])
        else:
            u = np.random.rand(*(lead + [N_samples]))
        u = torch.Tensor(np.ascontiguousarray(u)).to(bins.device).reshape(-1, N_samples)
    out = ops.sample_pdf(bins.reshape(-1, bins.shape[-1]), weights.reshape(-1, weights.shape[-1]), N_samples,
                         det=det, u=u, seed=int(torch.randint(0, 2 ** 31 - 1, (1,)).item()))
    return out.reshape(lead + [N_samples])


def compute_ssim(img0, img1, max_val=1.0, filter_size=11, filter_sigma=1.5, k1=0.01, k2=0.03, return_map=False):
    """Mean SSIM of image pairs `[..., width, height, channels]` (run_nerf_helpers.py:158-234: Gaussian-window
    SSIM with zero padding, variances clipped at 0 and the covariance clipped to sqrt(var0 var1)); with
    `return_map` the `[B, C, width, height]` map.  Evaluation-side helper of render_path, not a hot-path op."""
    x0 = torch.as_tensor(img0)
    x1 = torch.as_tensor(img1).to(x0.device)
    wd, ht, ch = x0.shape[-3:]
    # one plane per (batch, channel): [B*C, 1, wd, ht]
    p0 = x0.reshape(-1, wd, ht, ch).permute(0, 3, 1, 2).reshape(-1, 1, wd, ht)
    p1 = x1.reshape(-1, wd, ht, ch).permute(0, 3, 1, 2).reshape(-1, 1, wd, ht)
    half = filter_size // 2
    centre = (2 * half - filter_size + 1) / 2
    taps = torch.exp(-0.5 * ((torch.arange(filter_size, device=x0.device) - half + centre) / filter_sigma) ** 2)
    taps = taps / taps.sum()
    conv = torch.nn.functional.conv2d

    def blur(z):   # separable: along height first, then along width (zero padded)
        z = conv(z, taps.view(1, 1, 1, -1), padding=(0, half))
        return conv(z, taps.view(1, 1, -1, 1), padding=(half, 0))

    m0, m1 = blur(p0), blur(p1)
    v0 = (blur(p0 * p0) - m0 * m0).clamp(min=0.0)
    v1 = (blur(p1 * p1) - m1 * m1).clamp(min=0.0)
    cov = blur(p0 * p1) - m0 * m1
    cov = torch.sign(cov) * torch.minimum(torch.sqrt(v0 * v1), cov.abs())
    c1, c2 = (k1 * max_val) ** 2, (k2 * max_val) ** 2
    smap = ((2 * m0 * m1 + c1) * (2 * cov + c2)) / ((m0 * m0 + m1 * m1 + c1) * (v0 + v1 + c2))
    smap = smap.reshape(-1, ch, wd, ht)
    return smap if return_map else smap.reshape(smap.shape[0], -1).mean(-1)
