"""Synthetic H&E tiles generated on the device (SURVEY 8d recipe; torch RNG, not numpy's).

Benchmark / test input generator -- deliberately NOT part of the stainlib_amd package."""
import torch


def synth_tiles(n, h, w, seed=0, device="cuda", M_true=None, chunk=32):
    """Synthetic H&E tiles generated on the device (SURVEY 8d recipe; torch RNG, not numpy's)."""
    g = torch.Generator(device=device)
    g.manual_seed(int(seed))
    M = torch.tensor(M_true if M_true is not None else [[0.65, 0.70, 0.29], [0.07, 0.99, 0.11]],
                     dtype=torch.float32, device=device)
    M = M / M.norm(dim=1, keepdim=True)
    out = torch.empty((n, h, w, 3), dtype=torch.uint8, device=device)
    P = h * w
    for i in range(0, n, chunk):
        m = min(chunk, n - i)
        # Gamma(k=2, theta=0.35) = -0.35 * (ln U1 + ln U2)
        u = torch.rand((m, P, 2, 2), generator=g, device=device).clamp_min_(1e-12)
        Cc = -0.35 * (u[..., 0].log() + u[..., 1].log())
        bg = torch.rand((m, P, 1), generator=g, device=device) < 0.2
        Cc = torch.where(bg, Cc * 0.02, Cc)
        od = Cc @ M + 0.01 * torch.randn((m, P, 3), generator=g, device=device)
        rgb = (255.0 * torch.exp(-od)).clamp_(0, 255)
        out[i:i + m] = rgb.to(torch.uint8).reshape(m, h, w, 3)
        del u, Cc, bg, od, rgb
    return out
