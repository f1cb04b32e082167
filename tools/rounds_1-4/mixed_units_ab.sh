mkdir -p gpurun_out/r04_e2e
FIX=tests/golden/groth16_verify_compressed_1pub_golden.json
REST="pairing::double_in_place_circuit_montgomery,pairing::add_in_place_montgomery,pairing::mul_by_char_montgomery,bigint::multiplexer,g1::add_montgomery,inverse_iteration,inverse::divide_result_by_2^k::chunk,inverse::divide_result_by_even_part::chunk,fp254::exp_chunk"
export CR_REPS=1
out=gpurun_out/r04_e2e/verifier_mixed_units.log
echo "== M0: Fq12-level units (bench.py's plan)" > $out
python tools/rounds_1-4/concurrency_rate.py $FIX "fq12::square_montgomery,fq12::mul_montgomery,fq12::cyclotomic_square_montgomery,fq12::mul_by_034_montgomery,pairing::ell_by_constant_montgomery,$REST" 1,16 0 >> $out 2>&1
echo "== M1: fq12 mul/square as fq6::mul units" >> $out
python tools/rounds_1-4/concurrency_rate.py $FIX "fq6::mul_montgomery,fq12::cyclotomic_square_montgomery,fq12::mul_by_034_montgomery,pairing::ell_by_constant_montgomery,$REST" 1,16 0 >> $out 2>&1
cat $out
