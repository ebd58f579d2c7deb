"""One fitted model on several GPUs of ONE process, behind the single-handle interface (SURVEY section 8b's `n_devices`).

north_star partitions only the prediction batch: the fit runs on the first device, the model blob (header, scaled sources,
alpha, packed L^-1 — 286 MB at N = 8192) is copied device to device (`gpt_factor_copy`, hipMemcpyPeerAsync over xGMI: no
process group, no RCCL — a single process has no rendezvous to make; the multi-process form with one RCCL broadcast is
`distributed.broadcast_model`), and `predict_all` shards the query rows with `shard_range` onto one host thread per device
(ctypes releases the GIL inside the C call) writing into ONE set of output arrays.  Everything that is not a prediction
(export, return_cov, LML, timings) stays with the first handle.  `GaussianProcess(devices=[...])` is the user surface; the
reference's caller (transportation/gaussian_process_transportation.py:19-26) never sees the devices.

Unmeasured on multi-GPU hardware in this repository's rounds (the pool offers one GPU): rehearsed with the same device
listed twice — two handles, two streams, one GPU — against the single-handle results."""
from __future__ import annotations

from concurrent.futures import ThreadPoolExecutor

import numpy as np

from . import _lib
from .distributed import shard_range

_MIN_ROWS_PER_DEVICE = 256      # below this a shard is not worth a thread hand-off: the first device takes the batch


class DeviceGroup:
    def __init__(self, devices):
        devices = [int(d) for d in devices]
        if not devices:
            raise ValueError("devices must name at least one GPU")
        self.devices = devices
        self.handles = [_lib.Handle(d) for d in devices]
        self.primary = self.handles[0]
        self.device = devices[0]
        self._pool = ThreadPoolExecutor(max_workers=len(devices), thread_name_prefix="gpt-dev") if len(devices) > 1 else None
        self._replicated = False

    # ---- everything that is not a sharded prediction: the first handle
    def __getattr__(self, name):
        return getattr(self.primary, name)

    def close(self):
        if self._pool is not None:
            self._pool.shutdown(wait=True)
            self._pool = None
        for h in self.handles:
            h.close()

    # ---- fits: first device, then one device-to-device copy of the model per further device
    def _replicate(self):
        for h in self.handles[1:]:
            h.factor_copy_from(self.primary)
        self._replicated = True

    def fit(self, *a, **k):
        self._replicated = False
        self.primary.fit(*a, **k)
        self._replicate()

    def fit_noise_matrix(self, *a, **k):
        self._replicated = False
        self.primary.fit_noise_matrix(*a, **k)
        self._replicate()

    def fit_svgp(self, *a, **k):
        self._replicated = False
        self.primary.fit_svgp(*a, **k)
        self._replicate()

    def lml_objective(self, *a, **k):
        self._replicated = False            # leaves no model behind (gpt_lml_objective)
        return self.primary.lml_objective(*a, **k)

    # ---- sharded prediction
    def shards(self, M):
        """[(handle, start, stop)] — contiguous, balanced row ranges, one per device that gets any rows."""
        n = len(self.handles)
        if n == 1 or M < n * _MIN_ROWS_PER_DEVICE:
            return [(self.primary, 0, M)]
        return [(h, *shard_range(M, r, n)) for r, h in enumerate(self.handles)]

    def predict_all(self, Xq, mean=False, var=False, J=False, Jvar=False, dvar=False):
        if not self._replicated and len(self.handles) > 1:
            raise _lib.GptError("DeviceGroup: no fitted model on the devices (fit first)")
        flags = dict(mean=mean, var=var, J=J, Jvar=Jvar, dvar=dvar)
        Xq = np.asarray(Xq)
        parts = self.shards(Xq.shape[0] if Xq.ndim == 2 else 0)
        if len(parts) == 1:
            return self.primary.predict_all(Xq, **flags)
        nt, dt = self.primary.model_info()
        Xq = _lib.as_f64(Xq, 2, "X", dtype=_lib._NP_DTYPE[dt])       # validated once, sliced below without copies
        futs = [self._pool.submit(h.predict_all, Xq[a:b], **flags) for h, a, b in parts]
        outs = [f.result() for f in futs]                            # (an exception of any shard propagates)
        res = {}
        for key in ("mean", "var", "J", "Jvar"):
            res[key] = None if outs[0][key] is None else np.concatenate([o[key] for o in outs], axis=0)
        res["dvar"] = None if outs[0]["dvar"] is None else np.concatenate([o["dvar"] for o in outs], axis=1)   # (D, M)
        return res
