"""Device rate of a plan as a function of the instance count and of the number of calls that may run side by side (diagnostic tool).

  python tools/rounds_1-4/concurrency_rate.py <circuit spec> <units csv|coarse|fine> [instances csv] [concurrency csv]

Garbles with the ciphertexts kept on the device (retain) when they fit, else discarded window by window; prints gates/s and the
schedule's shape (batches, widest batch, depth in device steps)."""
import sys
import time

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))))
import garbled_snark_verifier_amd as gsv  # noqa: E402

COARSE = ["fq12::square_montgomery", "fq12::mul_montgomery", "fq12::cyclotomic_square_montgomery", "fq12::mul_by_034_montgomery", "pairing::ell_by_constant_montgomery",
          "pairing::double_in_place_circuit_montgomery", "pairing::add_in_place_montgomery", "pairing::mul_by_char_montgomery", "bigint::multiplexer", "g1::add_montgomery",
          "inverse_iteration", "inverse::divide_result_by_2^k::chunk", "inverse::divide_result_by_even_part::chunk", "fp254::exp_chunk"]
FINE = ["fq2::mul_montgomery", "fq2::square_montgomery", "fp254::mul_by_constant_montgomery", "bigint::mul_karatsuba", "fp254::montgomery_reduce", "bigint::multiplexer",
        "inverse_iteration", "inverse::divide_result_by_2^k::chunk", "inverse::divide_result_by_even_part::chunk"]


def main():
    spec = sys.argv[1]
    if spec.endswith(".json"):  # a fixture of tests/golden: its circuit name (carries the verifying key)
        import json
        spec = json.load(open(spec))["circuit"]
    units = {"coarse": COARSE, "fine": FINE}.get(sys.argv[2], sys.argv[2].split(","))
    insts = [int(x) for x in (sys.argv[3] if len(sys.argv) > 3 else "1,16").split(",")]
    concs = [int(x) for x in (sys.argv[4] if len(sys.argv) > 4 else "1,0").split(",")]
    eng = gsv.Engine(0)
    t0 = time.time()
    plan = gsv.Plan.from_circuit(spec, units, window_div=4 if max(insts) > 512 else None)
    print("plan: %d calls, %.3e gates, %.3e ciphertexts, built in %.1f s" % (plan.info["n_calls"], plan.info["n_gates"], plan.info["n_ciphertexts"], time.time() - t0), flush=True)
    n_in = plan.info["n_inputs"]
    for B in insts:
        labs = [gsv.labels_from_seed(100 + i, n_in) for i in range(B)]
        delta = np.stack([x[0] for x in labs]); consts = np.stack([np.stack([x[1], x[2]]) for x in labs]); inputs = np.stack([x[3] for x in labs])
        ref = None
        for conc in concs:
            retain = plan.info["n_ciphertexts"] * 16 * B < 60e9
            sess = gsv.Session(eng, plan, B, retain_stream=retain, concurrent_calls=conc)
            info = sess.schedule_info()
            best = None
            for _ in range(int(__import__("os").environ.get("CR_REPS", "3"))):
                sess.set_garble_inputs(delta, consts, inputs)
                t = time.time()
                if retain:
                    sess.garble(0); sess.sync()
                else:
                    sess.garble_streaming(discard=True)
                dt = time.time() - t
                best = dt if best is None else min(best, dt)
            out = sess.read_outputs()
            if ref is None:
                ref = out
            assert (out == ref).all(), "outputs differ between schedules"
            print("B=%4d conc=%3d (ni %d): %8.1f ms  %.3e gates/s | windows %d deps %d width %d depth %d of %d steps, scratch %.1f MB/inst, wire file %.1f MB/inst" % (
                B, conc, sess.instances_per_workgroup, best * 1e3, plan.info["n_gates"] * B / best, info["n_windows"], info["n_dependencies"], info["max_width"], info["critical_steps"], info["total_steps"],
                info["scratch_slots"] * 16 / 1e6, info["wire_file_slots"] * 16 / 1e6), flush=True)
            sess.close()
    plan.close()


if __name__ == "__main__":
    main()
