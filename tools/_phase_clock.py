import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import garbled_snark_verifier_amd as gsv
spec = sys.argv[1] if len(sys.argv) > 1 else "fq12_sqmul"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 512
eng = gsv.Engine(0)
prog = gsv.Program.from_circuit(spec, chain_feedback=True)
d, f, t, inp = gsv.labels_from_seed(1, prog.info["n_inputs"])
sess = gsv.Session(eng, prog, B, 3, 1)
sess.enable_step_clock()
sess.set_garble_inputs(np.tile(d, (B, 1)), np.tile(np.stack([f, t]), (B, 1, 1)), np.tile(inp, (B, 1, 1)))
for _ in range(2):
    sess.garble(0); sess.sync()
print("%s B=%d: %.2f ms/replay" % (spec, B, sess.last_kernel_ms() / 3))
c = sess.read_step_clock().astype(np.int64)[:160].reshape(16, 10)
names = ["and:load+wait", "and:aes", "and:store", "and passes", "xor phase", "barrier", "narrow steps", "n narrow", "wide pre-xor total", "total"]
for w in (0, 1, 7, 8, 15):
    tot = c[w, 9]
    print("wave %2d: total %d clk | " % (w, tot) + ", ".join("%s %.1f%%" % (names[i], 100.0 * c[w, i] / tot) for i in (0, 1, 2, 4, 5, 6)) + " | passes %d, clk/pass load %.0f aes %.0f store %.0f | narrow steps %d, %.0f clk each" % (
        c[w, 3], c[w, 0] / max(1, c[w, 3]), c[w, 1] / max(1, c[w, 3]), c[w, 2] / max(1, c[w, 3]), c[w, 7], c[w, 6] / max(1, c[w, 7])))
