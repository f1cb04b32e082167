FIX=tests/golden/groth16_verify_compressed_1pub_golden.json
REST="pairing::double_in_place_circuit_montgomery,pairing::add_in_place_montgomery,pairing::mul_by_char_montgomery,bigint::multiplexer,g1::add_montgomery,inverse_iteration,inverse::divide_result_by_2^k::chunk,inverse::divide_result_by_even_part::chunk,fp254::exp_chunk"
export CR_REPS=1
echo "== M1: fq12 mul/square as fq6::mul units" 
python tools/concurrency_rate.py $FIX "fq6::mul_montgomery,fq12::cyclotomic_square_montgomery,fq12::mul_by_034_montgomery,pairing::ell_by_constant_montgomery,$REST" 1,16 0
echo "== M2: every Fq12-level unit as Fq6-level units"
python tools/concurrency_rate.py $FIX "fq6::mul_montgomery,fq6::mul_by_fq2_montgomery,fq6::mul_by_01_montgomery,fq6::mul_by_01_constant1_montgomery,fq2::mul_montgomery,fq2::square_montgomery,$REST" 1,16 0
