"""Error map of the board kernel on one layer: max |err| per (16-channel row tile, 16-pixel column tile)."""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from sayuri_amd import _lib
from test_gpu_layers import conv_ref, _fp

def run(bsz, cin, cout, act, with_res, with_bias, seed=0):
    rng = np.random.default_rng(seed)
    n = len(bsz)
    xs = [rng.standard_normal((cin, b * b)).astype(np.float32) for b in bsz]
    w = (rng.standard_normal((cout, cin, 3, 3)) / np.sqrt(9 * cin)).astype(np.float32)
    bias = (rng.standard_normal(cout) * 0.1).astype(np.float32) if with_bias else np.zeros(cout, np.float32)
    res = [rng.standard_normal((cout, b * b)).astype(np.float32) for b in bsz] if with_res else None
    xs_r = [x.astype(np.float16).astype(np.float64) for x in xs]
    w_r = w.astype(np.float16).astype(np.float64)
    res_r = [r.astype(np.float16).astype(np.float64) for r in res] if res else None
    ref = conv_ref(xs_r, bsz, w_r, bias.astype(np.float64), res_r, 3, False, act, False)
    xcat = np.concatenate([x.ravel() for x in xs])
    rcat = np.concatenate([r.ravel() for r in res]) if res else None
    y = np.zeros(sum(cout * b * b for b in bsz), np.float32)
    bs_arr = np.asarray(bsz, np.int32)
    lib = _lib.hip()
    rc = lib.sayuri_hip_test_conv(0, 1, n, bs_arr.ctypes.data_as(_lib.c_int_p), 19, cin, cout, 3, 0, act, 0, _fp(xcat), _fp(w.ravel()),
                                  _fp(bias), _fp(rcat) if res else None, _fp(y))
    assert rc == 0, lib.sayuri_hip_last_error().decode()
    print(f"case bsz={bsz} {cin}->{cout} act={act} res={with_res} bias={with_bias} kind={lib.sayuri_hip_test_last_conv_kind()}")
    off = 0
    for i, b in enumerate(bsz):
        got = y[off:off + cout * b * b].reshape(cout, b * b); off += cout * b * b
        err = np.abs(got - ref[i])
        print(f" sample {i}: max err {err.max():.4f} scale {np.abs(ref[i]).max():.3f}")
        if err.max() > 4e-3 * np.abs(ref[i]).max():
            ncol = (b * b + 15) // 16
            m = np.zeros((cout // 16, ncol))
            for rt in range(cout // 16):
                for ct in range(ncol):
                    m[rt, ct] = err[rt * 16:(rt + 1) * 16, ct * 16:(ct + 1) * 16].max()
            np.set_printoptions(linewidth=250, precision=1, suppress=True)
            print((m > 4e-3 * np.abs(ref[i]).max()).astype(int))
            # inside the first bad tile
            rt, ct = np.argwhere(m > 4e-3 * np.abs(ref[i]).max())[0]
            e = err[rt * 16:(rt + 1) * 16, ct * 16:(ct + 1) * 16]
            print(f"  first bad tile rt={rt} ct={ct}: per-channel(row) x per-pixel(col) bad mask")
            print((e > 4e-3 * np.abs(ref[i]).max()).astype(int))
            return False
    return True

if __name__ == "__main__":
    ok = True
    for args in [([19], 32, 128, 0, False, False), ([19], 256, 256, 0, False, False), ([19], 256, 256, 0, False, True),
                 ([19], 256, 256, 0, True, True), ([19], 256, 256, 5, True, True), ([19] * 3, 256, 256, 5, True, True),
                 ([13] * 5, 128, 192, 0, False, False), ([9] * 9, 64, 128, 0, False, False)]:
        ok &= run(*args)
    print("ALL OK" if ok else "FAILURES")
