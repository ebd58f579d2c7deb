"""Five consecutive fits at N = 8192 on the handle's own stream, on torch's current (null) stream and on a torch side stream:
does the caller's stream change the fit (CU-masked streams + events inside)?  No: 9.45-9.55 ms from the third call on in
all three; the second call is still 4 % slow.  usage: python tools/fit_stream_probe.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussian_process_transportation_amd import _lib  # noqa: E402

N = 8192
rng = np.random.default_rng(0)
X = rng.uniform(0, 1, (N, 3)); Y = np.sin(4 * X); ls = np.array([0.1] * 3)
for mode in ("own stream", "torch current (null) stream", "torch side stream"):
    h = _lib.Handle(0)
    if mode.startswith("torch current"):
        h.set_stream(torch.cuda.current_stream().cuda_stream)
    elif mode.startswith("torch side"):
        st = torch.cuda.Stream(); h.set_stream(st.cuda_stream)
    ts = []
    for rep in range(5):
        h.fit(X, Y, ls, 0.1, 1e-4, 1e-10)
        ts.append(h.fit_timings()["total"])
    print(mode, " ".join(f"{t:.2f}" for t in ts), flush=True)
    h.close()
