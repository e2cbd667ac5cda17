mkdir -p gpurun_out/r02x
B="--backend hip --num-cols 65536 --num-runs 15 --input synthetic"
for t in "" "--tune-placement"; do
python -m gt4py_dwarf_p_cloudsc2_tl_ad_amd.drivers.run_taylor_test $B $t 2>&1 | grep -E "completed|placement tuned|The test (passed|failed)" | sed "s/^/taylor $t: /"
python -m gt4py_dwarf_p_cloudsc2_tl_ad_amd.drivers.run_taylor_test $B --fused $t 2>&1 | grep -E "completed|placement tuned" | sed "s/^/taylor --fused $t: /"
python -m gt4py_dwarf_p_cloudsc2_tl_ad_amd.drivers.run_symmetry_test $B $t 2>&1 | grep -E "completed|placement tuned|symmetry test" | sed "s/^/symmetry $t: /"
done
