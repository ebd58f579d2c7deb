"""Several handles driven from several threads at once (what the concurrent restart runs of the hyper-parameter search do),
with different problem shapes per thread: every result must be bit-identical to the same call made alone.
usage: python tools/thread_stress.py [seconds]"""
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gaussian_process_transportation_amd import _lib  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
shapes = [(300, 2, 2, 500), (700, 3, 3, 3000), (1100, 3, 1, 460), (1500, 5, 2, 900), (520, 8, 3, 200), (2500, 3, 3, 1000), (64, 1, 1, 70), (900, 6, 6, 4100)]


def work(h, N, D, O, M, seed):
    rng = np.random.default_rng(seed)
    X = rng.uniform(0, 1, (N, D)); Y = np.sin(3 * X[:, :1] + np.arange(O)[None, :]); Xq = rng.uniform(0, 1, (M, D))
    ls = np.full(D, 0.2 if D <= 3 else 0.6)
    v, g = h.lml_objective(X, Y, ls, 0.5, 1e-3, 1e-10)
    h.fit(X, Y, ls, 0.5, 1e-3, 1e-10)
    out = h.predict_all(Xq, mean=True, var=True, J=True, Jvar=True, dvar=True)
    return [np.float64(v), g] + [out[k] for k in ("mean", "var", "J", "Jvar", "dvar")]


h0 = _lib.Handle(0)
alone = [work(h0, *s, seed=i) for i, s in enumerate(shapes)]
h0.close()
errors = []
rounds = [0] * len(shapes)


def thread(i):
    h = _lib.Handle(0)
    t0 = time.time()
    try:
        while time.time() - t0 < budget:
            got = work(h, *shapes[i], seed=i)
            for a, b in zip(got, alone[i]):
                if not np.array_equal(a, b):
                    errors.append((i, shapes[i]))
                    return
            rounds[i] += 1
    except Exception as e:  # noqa: BLE001
        errors.append((i, repr(e)))
    finally:
        h.close()


ts = [threading.Thread(target=thread, args=(i,)) for i in range(len(shapes))]
for t in ts:
    t.start()
while any(t.is_alive() for t in ts):
    time.sleep(10)
    print("... rounds per thread", rounds, flush=True)
for t in ts:
    t.join()
print(f"{len(shapes)} threads, {sum(rounds)} rounds, mismatches / errors: {errors}")
sys.exit(1 if errors else 0)
