"""`KalmanFilter` with the reference's call surface (demo_bottrack_onnx_tflite.py:118-380), every
method running as a HIP kernel on the MI355X through the C ABI (csrc/kalman.hip).

dtype flow follows the reference under NumPy 2: `initiate` on a float32
measurement returns float32 arrays; everything else returns float64.  A float32 `mean` handed to
`predict`/`multi_predict`/`project`/`update` makes the kernels evaluate the noise terms in float32
exactly as NumPy scalar promotion does for float32 state.
"""
from __future__ import annotations

import numpy as np

from . import _lib


def _dev(a, dtype):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a, dtype=dtype)).cuda()


class KalmanFilter:
    chi2inv95 = {1: 3.8415, 2: 5.9915, 3: 7.8147, 4: 9.4877, 5: 11.070,
                 6: 12.592, 7: 14.067, 8: 15.507, 9: 16.919}          # D:139-149

    def __init__(self, context: "_lib.Context | None" = None):
        ndim, dt = 4, 1.0
        self._motion_mat = np.eye(2 * ndim, 2 * ndim)                 # D:151-158 (kept for API parity;
        for i in range(ndim):                                         #  the kernels never form F or H)
            self._motion_mat[i, ndim + i] = dt
        self._update_mat = np.eye(ndim, 2 * ndim)
        self._std_weight_position = 1.0 / 20
        self._std_weight_velocity = 1.0 / 160
        self._ctx = context

    @property
    def ctx(self):
        if self._ctx is None:
            self._ctx = _lib.default_context()
        return self._ctx

    # D:166-197
    def initiate(self, measurement):
        import torch
        lib = _lib.load(require_gpu=True)
        z = _dev(np.asarray(measurement).reshape(1, 4), np.float32)
        mean = torch.empty((1, 8), dtype=torch.float64, device="cuda")
        cov = torch.empty((1, 8, 8), dtype=torch.float64, device="cuda")
        _lib.check(lib.bts_kf_initiate(self.ctx.handle, _lib.ptr(z), 1, _lib.ptr(mean), _lib.ptr(cov),
                                       _lib.current_stream()))
        return mean[0].cpu().numpy().astype(np.float32), cov[0].cpu().numpy().astype(np.float32)

    # D:199-234
    def predict(self, mean, covariance):
        m, c = self.multi_predict(np.asarray(mean)[None], np.asarray(covariance)[None])
        return m[0], c[0]

    # D:236-263
    def project(self, mean, covariance):
        import torch
        lib = _lib.load(require_gpu=True)
        f32 = np.asarray(mean).dtype == np.float32
        m = _dev(np.asarray(mean).reshape(1, 8), np.float64)
        c = _dev(np.asarray(covariance).reshape(1, 8, 8), np.float64)
        flag = torch.full((1,), 1 if f32 else 0, dtype=torch.uint8, device="cuda")
        pm = torch.empty((1, 4), dtype=torch.float64, device="cuda")
        pc = torch.empty((1, 4, 4), dtype=torch.float64, device="cuda")
        _lib.check(lib.bts_kf_project(self.ctx.handle, _lib.ptr(m), _lib.ptr(c), _lib.ptr(flag), 1,
                                      _lib.ptr(pm), _lib.ptr(pc), _lib.current_stream()))
        return pm[0].cpu().numpy(), pc[0].cpu().numpy()

    # D:265-302
    def multi_predict(self, mean, covariance, zero_velocity=None):
        """`zero_velocity` (bool[N], optional) applies STrack.multi_predict's rule for
        non-Tracked rows (D:529-532) inside the same kernel."""
        import torch
        lib = _lib.load(require_gpu=True)
        mean = np.asarray(mean)
        n = len(mean)
        f32 = mean.dtype == np.float32
        m = _dev(mean.reshape(n, 8), np.float64)
        c = _dev(np.asarray(covariance).reshape(n, 8, 8), np.float64)
        flag = torch.full((max(n, 1),), 1 if f32 else 0, dtype=torch.uint8, device="cuda")
        zv = None if zero_velocity is None else _dev(np.asarray(zero_velocity).astype(np.uint8), np.uint8)
        _lib.check(lib.bts_kf_multi_predict(self.ctx.handle, _lib.ptr(m), _lib.ptr(c), _lib.ptr(zv),
                                            _lib.ptr(flag), n, None, _lib.current_stream()))
        return m.cpu().numpy(), c.cpu().numpy()

    # D:304-336
    def update(self, mean, covariance, measurement):
        import torch
        lib = _lib.load(require_gpu=True)
        f32 = np.asarray(mean).dtype == np.float32
        m = _dev(np.asarray(mean).reshape(1, 8), np.float64)
        c = _dev(np.asarray(covariance).reshape(1, 8, 8), np.float64)
        z = _dev(np.asarray(measurement).reshape(1, 4), np.float32)
        flag = torch.full((1,), 1 if f32 else 0, dtype=torch.uint8, device="cuda")
        _lib.check(lib.bts_kf_update(self.ctx.handle, _lib.ptr(m), _lib.ptr(c), _lib.ptr(flag), None,
                                     _lib.ptr(z), None, 1, _lib.current_stream()))
        return m[0].cpu().numpy(), c[0].cpu().numpy()

    def update_batch(self, mean, covariance, measurements, trk_idx=None, meas_idx=None):
        """Batched form used by the tracker: rows trk_idx[q] corrected with measurements[meas_idx[q]]."""
        import torch
        lib = _lib.load(require_gpu=True)
        mean = np.asarray(mean)
        f32 = mean.dtype == np.float32
        n = len(mean)
        m = _dev(mean.reshape(n, 8), np.float64)
        c = _dev(np.asarray(covariance).reshape(n, 8, 8), np.float64)
        z = _dev(np.asarray(measurements).reshape(-1, 4), np.float32)
        flag = torch.full((max(n, 1),), 1 if f32 else 0, dtype=torch.uint8, device="cuda")
        ti = None if trk_idx is None else _dev(trk_idx, np.int32)
        mi = None if meas_idx is None else _dev(meas_idx, np.int32)
        k = n if trk_idx is None else len(trk_idx)
        _lib.check(lib.bts_kf_update(self.ctx.handle, _lib.ptr(m), _lib.ptr(c), _lib.ptr(flag), _lib.ptr(ti),
                                     _lib.ptr(z), _lib.ptr(mi), k, _lib.current_stream()))
        return m.cpu().numpy(), c.cpu().numpy()

    def multi_gmc(self, mean, covariance, H):
        """STrack.multi_gmc's arithmetic (D:538-554) on arrays mean[N,8], cov[N,8,8]; H is 2x3."""
        lib = _lib.load(require_gpu=True)
        mean = np.asarray(mean)
        n = len(mean)
        m = _dev(mean.reshape(n, 8), np.float64)
        c = _dev(np.asarray(covariance).reshape(n, 8, 8), np.float64)
        h = np.ascontiguousarray(H, dtype=np.float64).reshape(6)
        _lib.check(lib.bts_kf_multi_gmc(self.ctx.handle, _lib.ptr(m), _lib.ptr(c), n, h.ctypes.data, _lib.current_stream()))
        return m.cpu().numpy(), c.cpu().numpy()

    # D:338-380
    def gating_distance(self, mean, covariance, measurements, only_position=False, metric="maha"):
        import torch
        lib = _lib.load(require_gpu=True)
        if metric not in ("maha", "gaussian"):
            raise ValueError("invalid distance metric")          # D:379-380
        mean = np.asarray(mean)
        single = mean.ndim == 1
        m = _dev(mean.reshape(-1, 8), np.float64)
        n = m.shape[0]
        c = _dev(np.asarray(covariance).reshape(n, 8, 8), np.float64)
        z = _dev(np.asarray(measurements).reshape(-1, 4), np.float64)
        k = z.shape[0]
        if k == 0:
            return np.zeros((0,)) if single else np.zeros((n, 0))
        out = torch.empty((n, k), dtype=torch.float64, device="cuda")
        _lib.check(lib.bts_kf_gating(self.ctx.handle, _lib.ptr(m), _lib.ptr(c), n, _lib.ptr(z), k,
                                     1 if only_position else 0, 0 if metric == "maha" else 1, _lib.ptr(out),
                                     _lib.current_stream()))
        res = out.cpu().numpy()
        return res[0] if single else res
