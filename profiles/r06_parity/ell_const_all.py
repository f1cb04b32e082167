import sys, time
sys.path.insert(0, '/root/repo/tests'); sys.path.insert(0, '/root/repo')
import numpy as np
import ref_gadgets as R, ref_stream_compare as S
ks = list(range(91))
for k in ks:
    name = "ell_const:%d" % k
    R.CIRCUITS[name] = (3048 + 762, R._ell_const_circuit(k))
    t0 = time.time()
    exp = S.restated_stream(name)
    t, a, b, c, ins, outs = S.product_stream(name, 40_000_000)
    got = S.canonical_np(t, a, b, c, ins, outs[:len(exp[4])], dead_marker=0xFFFFFFFF)
    j = S.first_difference(got, exp)
    ok = j is None and len(got[0]) == len(exp[0]) and (got[4] == exp[4]).all()
    print("%s: %d gates (%d dead) %s (%.0f s)" % (name, len(got[0]), int(got[3].sum()), "identical" if ok else "DIFFERENT at gate %s" % j, time.time() - t0), flush=True)
