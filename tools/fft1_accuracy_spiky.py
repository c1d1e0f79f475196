import sys, numpy as np, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from ipde_amd.spectral import fft1
from test_geometry_cpu import _density_with_noise_turnaround
for n in (9552, 9560, 2390):
    rng = np.random.default_rng(n)
    _density_with_noise_turnaround(n, rng, n // 8, 0.5)
    mu = _density_with_noise_turnaround(n, rng, n // 5, 20.0)
    z = mu[:n] + 1j * mu[n:]
    for batch in (1, 14):
        zz = np.tile(z, (batch, 1))
        Z = fft1(torch.as_tensor(zz, device='cuda'), -1).cpu().numpy()[0]
        Zn = np.fft.fft(z)
        back = fft1(torch.as_tensor(np.tile(Zn, (batch, 1)), device='cuda'), +1).cpu().numpy()[0]
        print(n, batch, 'forward max err / max|Z| %.2e' % (np.abs(Z - Zn).max() / np.abs(Zn).max()),
              ' low modes (k < 200): %.2e' % (np.abs(Z - Zn)[:200].max() / np.abs(Zn).max()),
              ' inverse err / max %.2e' % (np.abs(back - z).max() / np.abs(z).max()))
