"""Hand-written five-pass FFT Poisson solve (fftpm.hip) against the float64 oracle, and its adjoint."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import pm_oracle as o  # noqa: E402  (checker only)


def rel_l2(a, b):
    dt = np.complex128 if (np.iscomplexobj(a) or np.iscomplexobj(b)) else np.float64
    a, b = np.asarray(a, dt), np.asarray(b, dt)
    return float(np.linalg.norm(a - b) / np.linalg.norm(b))


@pytest.fixture(scope="module")
def nb(gpu):
    from montecosmo_amd import nbody
    return nbody


def _force_meshes(nb, rho):
    import torch
    plan = nb.get_plan(rho.shape)
    r = torch.from_numpy(rho).cuda()
    fm = torch.empty((3,) + rho.shape, dtype=torch.float32, device="cuda")
    plan.call("mcpm_force_meshes_f32", C.c_void_p(r.data_ptr()), C.c_void_p(fm.data_ptr()))
    return fm.cpu().numpy()


def _force_meshes_vjp(nb, fbar):
    import torch
    shape = fbar.shape[1:]
    plan = nb.get_plan(shape)
    f = torch.from_numpy(fbar).cuda()
    out = torch.empty(shape, dtype=torch.float32, device="cuda")
    plan.call("mcpm_force_meshes_vjp_f32", C.c_void_p(f.data_ptr()), C.c_void_p(out.data_ptr()))
    return out.cpu().numpy()


@pytest.mark.parametrize("shape", [(64, 64, 64), (128, 64, 256), (256, 128, 64), (64, 512, 128), (512, 64, 64),
                                   (64, 64, 1024), (1024, 64, 64), (64, 1024, 64), (48, 32, 16)])
def test_force_meshes_match_oracle(nb, shape):
    rng = np.random.default_rng(0)
    rho = rng.standard_normal(shape).astype(np.float32)
    got = _force_meshes(nb, rho)
    want = o.force_meshes(np.fft.rfftn(rho.astype(np.float64)))
    for c in range(3):
        assert rel_l2(got[c], want[c]) < 2e-6, c


@pytest.mark.parametrize("shape", [(64, 64, 64), (128, 256, 64), (64, 128, 512), (48, 32, 16)])
def test_force_meshes_vjp_is_the_adjoint(nb, shape):
    rng = np.random.default_rng(1)
    fbar = rng.standard_normal((3,) + shape).astype(np.float32)
    got = _force_meshes_vjp(nb, fbar)
    # oracle adjoint: rho_bar = irfftn(sum_c conj(m_c) rfftn(f_bar_c))
    kvec = o.rfftk(shape)
    acc = 0
    for c in range(3):
        m = -o.gradient_hat(kvec, c) * o.invlaplace_hat(kvec)
        acc = acc + np.conj(m) * np.fft.rfftn(fbar[c].astype(np.float64))
    want = np.fft.irfftn(acc, s=shape, axes=(0, 1, 2))
    assert rel_l2(got, want) < 2e-6
    # dot test against the forward operator on the GPU
    rho = rng.standard_normal(shape).astype(np.float32)
    fm = _force_meshes(nb, rho)
    lhs = np.sum(fm.astype(np.float64) * fbar)
    rhs = np.sum(rho.astype(np.float64) * got)
    assert abs(lhs - rhs) < 1e-5 * np.sqrt(np.sum(fm.astype(np.float64) ** 2) * np.sum(fbar.astype(np.float64) ** 2))


@pytest.mark.parametrize("shape", [(64, 64, 64), (128, 64, 256), (48, 40, 24)])
def test_generic_rfftn_irfftn_numpy_semantics(gpu, shape):
    """nbody.rfftn / irfftn (hand-written passes on power-of-two meshes, rocFFT otherwise) against numpy, including a
    NON-Hermitian half-spectrum: numpy's irfftn (ifft over x, y, then c2r over z) silently projects it, and so must we
    (the interlacing phases of nufft violate Hermitian symmetry on the Nyquist planes, nbody.py:525)."""
    from montecosmo_amd import nbody
    rng = np.random.default_rng(12)
    x = rng.standard_normal(shape).astype(np.float32)
    X = nbody.rfftn(x).cpu().numpy()
    assert rel_l2(X, np.fft.rfftn(x.astype(np.float64))) < 2e-6
    Z = (rng.standard_normal(X.shape) + 1j * rng.standard_normal(X.shape)).astype(np.complex64)
    y = nbody.irfftn(Z).cpu().numpy()
    assert rel_l2(y, np.fft.irfftn(Z.astype(np.complex128), s=shape, axes=(0, 1, 2))) < 2e-6
