"""Device (torch) builders of the dense Laplace / Stokes layer forms used by the QFS
set-up — the same formulas as ipde_amd.pybie2d_compat.*_Layer_Form / *_Singular_Form
(which stay the host reference; tests compare the two).  At config-5 scale the 2N x 2N
Stokes forms (N = 12 400: 4.9 GB each, ~20 numpy passes over N^2 temporaries) were most of
the 51 s set-up; on the GPU each is a few milliseconds and no matrix crosses PCIe.
"""
import numpy as np
import torch

from .pybie2d_compat import _kress_log_weights


def _t(a, dev):
    return torch.as_tensor(np.ascontiguousarray(a, dtype=np.float64), device=dev)


def _geom(source, target, dev):
    dx = _t(target.x, dev)[:, None] - _t(source.x, dev)[None, :]
    dy = _t(target.y, dev)[:, None] - _t(source.y, dev)[None, :]
    return dx, dy


def laplace_form(source, target, dev, ifcharge=False, ifdipole=False):
    dx, dy = _geom(source, target, dev)
    d2 = dx * dx + dy * dy
    w = _t(source.weights, dev)[None, :]
    out = torch.zeros_like(d2)
    if ifcharge:
        out += (-0.25 / np.pi) * torch.log(d2) * w
    if ifdipole:
        nd = dx * _t(source.normal_x, dev)[None, :] + dy * _t(source.normal_y, dev)[None, :]
        out += (0.5 / np.pi) * nd / d2 * w
    return out


def _toeplitz_index(N, dev):
    i = torch.arange(N, device=dev)
    return (i[:, None] - i[None, :]).abs()


def laplace_singular_form(bdy, dev, ifcharge=False, ifdipole=False):
    N = bdy.N
    dx, dy = _geom(bdy, bdy, dev)
    d2 = dx * dx + dy * dy
    w = _t(bdy.weights, dev)[None, :]
    speed = _t(bdy.speed, dev)
    out = torch.zeros_like(d2)
    if ifcharge:
        sep = _toeplitz_index(N, dev)
        s2t = 4 * np.sin(0.5 * bdy.dt * np.arange(N)) ** 2
        s2t[0] = 1.0
        s2 = _t(s2t, dev)[sep]
        d2f = d2.clone()
        d2f.fill_diagonal_(1.0)
        smooth = (-0.25 / np.pi) * torch.log(d2f / s2)
        smooth.diagonal().copy_((-0.5 / np.pi) * torch.log(speed))
        R = _t(_kress_log_weights(N), dev)[sep]
        out += (-0.25 / np.pi) * R * speed[None, :] + smooth * w
    if ifdipole:
        d2f = d2.clone()
        d2f.fill_diagonal_(1.0)
        nd = dx * _t(bdy.normal_x, dev)[None, :] + dy * _t(bdy.normal_y, dev)[None, :]
        D = (0.5 / np.pi) * nd / d2f
        D.diagonal().copy_(_t(-bdy.curvature / (4 * np.pi), dev))
        out += D * w
    return out


def _blocks(Bxx, Bxy, Byy):
    return torch.cat([torch.cat([Bxx, Bxy], dim=1), torch.cat([Bxy, Byy], dim=1)], dim=0)


def stokes_form(source, target, dev, ifforce=False, ifdipole=False):
    dx, dy = _geom(source, target, dev)
    d2 = dx * dx + dy * dy
    id2 = 1.0 / d2
    w = _t(source.weights, dev)[None, :]
    Bxx, Bxy, Byy = torch.zeros_like(d2), torch.zeros_like(d2), torch.zeros_like(d2)
    if ifforce:
        c = 0.25 / np.pi
        lg = -0.5 * torch.log(d2)
        Bxx += c * (lg + dx * dx * id2) * w
        Bxy += c * (dx * dy * id2) * w
        Byy += c * (lg + dy * dy * id2) * w
    if ifdipole:
        q = (dx * _t(source.normal_x, dev)[None, :] + dy * _t(source.normal_y, dev)[None, :]) \
            * id2 * id2 * w / np.pi
        Bxx += q * dx * dx
        Bxy += q * dx * dy
        Byy += q * dy * dy
    return _blocks(Bxx, Bxy, Byy)


def stokes_singular_form(bdy, dev, ifforce=False, ifdipole=False):
    N = bdy.N
    dx, dy = _geom(bdy, bdy, dev)
    d2 = dx * dx + dy * dy
    d2.fill_diagonal_(1.0)
    id2 = 1.0 / d2
    w = _t(bdy.weights, dev)[None, :]
    tx, ty = _t(bdy.tangent_x, dev), _t(bdy.tangent_y, dev)
    Bxx = torch.zeros((N, N), dtype=torch.float64, device=dev)
    Bxy, Byy = torch.zeros_like(Bxx), torch.zeros_like(Bxx)
    if ifforce:
        c = 0.25 / np.pi
        half_slp = 0.5 * laplace_singular_form(bdy, dev, ifcharge=True)
        Rxx, Rxy, Ryy = dx * dx * id2, dx * dy * id2, dy * dy * id2
        Rxx.diagonal().copy_(tx * tx)
        Rxy.diagonal().copy_(tx * ty)
        Ryy.diagonal().copy_(ty * ty)
        Bxx += half_slp + c * Rxx * w
        Bxy += c * Rxy * w
        Byy += half_slp + c * Ryy * w
    if ifdipole:
        q = (dx * _t(bdy.normal_x, dev)[None, :] + dy * _t(bdy.normal_y, dev)[None, :]) * id2 * id2 / np.pi
        Dxx, Dxy, Dyy = q * dx * dx, q * dx * dy, q * dy * dy
        lim = _t(-bdy.curvature / (2 * np.pi), dev)
        Dxx.diagonal().copy_(lim * tx * tx)
        Dxy.diagonal().copy_(lim * tx * ty)
        Dyy.diagonal().copy_(lim * ty * ty)
        Bxx += Dxx * w
        Bxy += Dxy * w
        Byy += Dyy * w
    return _blocks(Bxx, Bxy, Byy)


def stokes_pressure_fix(source, target, dev):
    nt = torch.cat([_t(target.normal_x, dev), _t(target.normal_y, dev)])
    ws = _t(source.weights, dev)
    ns = torch.cat([_t(source.normal_x, dev) * ws, _t(source.normal_y, dev) * ws])
    return torch.outer(nt, ns) / float(np.sum(source.weights))
