"""Synthetic multi-emitter snapshot streams for bench.py / demos, generated on the GPU with torch
(uncorrelated unit-power complex-Gaussian emitters, AWGN, antenna-interleaved
in[c*m + r], complex64).  Independent of the test oracle."""
from __future__ import annotations

import math

import numpy as np

C_LIGHT = 299792458.0


def array_geometry(m):
    """m=4: unit square; otherwise uniform circle with unit adjacent spacing (units of array_spacing)."""
    if m == 4:
        return [[0.0, 0.0], [1.0, 0.0], [1.0, 1.0], [0.0, 1.0]]
    r = 0.5 / math.sin(math.pi / m)
    return [[r * math.cos(2 * math.pi * k / m), r * math.sin(2 * math.pi * k / m)] for k in range(m)]


def steering(theta_deg, antenna_array, array_spacing, wavelength):
    th = math.radians(theta_deg)
    p = np.asarray(antenna_array, dtype=np.float64) * array_spacing
    return np.exp(-2j * np.pi * (p[:, 0] * math.cos(th) + p[:, 1] * math.sin(th)) / wavelength)


def synth_stream(torch, device, batch, m, nsamples, antenna_array, frequency, array_spacing,
                 angles_deg=(40.3, 121.7), snr_db=20.0, seed=0):
    """(batch, nsamples) complex64 tensor on `device` (returned as a float32 view (batch, 2*nsamples))."""
    g = torch.Generator(device=device)
    g.manual_seed(int(seed))
    K = nsamples // m
    lam = C_LIGHT / frequency
    x = torch.zeros(batch, K, m, dtype=torch.complex64, device=device)
    for th in angles_deg:
        a = torch.from_numpy(steering(th, antenna_array, array_spacing, lam).astype(np.complex64)).to(device)
        s = torch.randn(batch, K, 2, generator=g, device=device, dtype=torch.float32)
        s = torch.view_as_complex(s) * (1.0 / math.sqrt(2.0))
        x += s[:, :, None] * a[None, None, :]
    sigma = 10.0 ** (-snr_db / 20.0) / math.sqrt(2.0)
    nz = torch.randn(batch, K, m, 2, generator=g, device=device, dtype=torch.float32)
    x += torch.view_as_complex(nz) * sigma
    return torch.view_as_real(x.reshape(batch, K * m)).reshape(batch, 2 * nsamples).contiguous()


def synth_scenes(torch, device, batch, m, nsamples, antenna_array, frequency, array_spacing, n_emitters=2, snr_db=20.0, seed=0):
    """Like synth_stream, but every ITEM sees its own scene: n_emitters angles drawn uniformly over 360 degrees per item
    (an incoherent batch -- items of many unrelated streams -- the unfavourable case for anything that is decided per
    wave of 16 items: the scan's top-n gate, the coarse-gated scan's tile votes)."""
    g = torch.Generator(device=device)
    g.manual_seed(int(seed))
    K = nsamples // m
    lam = C_LIGHT / frequency
    p = torch.tensor(np.asarray(antenna_array, dtype=np.float64) * array_spacing, device=device)         # (m, 2)
    x = torch.zeros(batch, K, m, dtype=torch.complex64, device=device)
    for _ in range(n_emitters):
        th = torch.rand(batch, generator=g, device=device, dtype=torch.float64) * (2.0 * math.pi)
        phase = (p[None, :, 0] * torch.cos(th)[:, None] + p[None, :, 1] * torch.sin(th)[:, None]) * (-2.0 * math.pi / lam)
        a = torch.polar(torch.ones_like(phase), phase).to(torch.complex64)                                # (batch, m)
        s = torch.randn(batch, K, 2, generator=g, device=device, dtype=torch.float32)
        s = torch.view_as_complex(s) * (1.0 / math.sqrt(2.0))
        x += s[:, :, None] * a[:, None, :]
    sigma = 10.0 ** (-snr_db / 20.0) / math.sqrt(2.0)
    nz = torch.randn(batch, K, m, 2, generator=g, device=device, dtype=torch.float32)
    x += torch.view_as_complex(nz) * sigma
    return torch.view_as_real(x.reshape(batch, K * m)).reshape(batch, 2 * nsamples).contiguous()
