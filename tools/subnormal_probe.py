import sys, os, zlib
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import numpy as np
from conftest import load_package
import oracle
from test_gpu_certificate import make_case, bf16_rne, exact_distances
vdb = load_package()
tau = 2.0 ** -126
for metric in (2, 0, 1):
    rng = np.random.default_rng(zlib.crc32(f"{metric}/subnormal".encode()))
    rows, q = make_case("subnormal", metric, rng)
    n, d = rows.shape
    ix = vdb.GpuFlatIndex(vdb.DistanceMetric(metric), keep_host_copy=False); ix.add_bulk(rows)
    s_raw, qinfo, c = ix.debug_screen_scores(q, raw=True)
    info = ix.debug_row_info().astype(np.float64)
    D = q.astype(np.float64) @ rows.astype(np.float64).T
    D16 = bf16_rne(q).astype(np.float64) @ bf16_rne(rows).astype(np.float64).T
    nd = np.linalg.norm(rows.astype(np.float64), axis=1); qn = np.linalg.norm(q.astype(np.float64), axis=1)
    if metric == 2:
        acc = -s_raw.astype(np.float64)
        e16 = np.abs(acc - D16)
        print("dot: max |acc-D16| =", e16.max(), " in units of K*tau:", e16.max() / (d * tau), " max |acc - D| / (K tau)", np.abs(acc - D).max() / (d * tau))
        # which pairs: relation to operand magnitudes
        i = np.unravel_index(np.argmax(e16), e16.shape)
        print(" at", i, "qn", qn[i[0]], "nd", nd[i[1]], "acc", acc[i], "D16", D16[i], "D", D[i], "max|q|", np.abs(q[i[0]]).max(), "max|d|", np.abs(rows[i[1]]).max())
        # fraction where acc == 0 but D16 != 0
        z = (acc == 0) & (D16 != 0)
        print(" acc==0 & D16!=0:", z.sum(), "largest |D16| among them", np.abs(D16[z]).max() if z.any() else 0, " (tau =", tau, ")")
        nz = (acc != 0)
        print(" smallest nonzero |acc|", np.abs(acc[nz]).min())
        # per-element products: is the error explained by flushing operands with |x| < tau?
        qf = np.where(np.abs(bf16_rne(q)) < tau, 0, bf16_rne(q)).astype(np.float64); rf = np.where(np.abs(bf16_rne(rows)) < tau, 0, bf16_rne(rows)).astype(np.float64)
        Df = qf @ rf.T
        print(" with operands below tau flushed: max |acc - Df| / (K tau) =", np.abs(acc - Df).max() / (d * tau))
