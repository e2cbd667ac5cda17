for P in double single; do for N in 100 1001 10000 65536 262144; do
python -m gt4py_dwarf_p_cloudsc2_tl_ad_amd.drivers.run_taylor_test --num-cols $N --num-runs 3 --precision $P --fused-all --graph 2>&1 | grep -E "^The test|Traceback|Error" | tr '\n' ' '; echo " <- taylor fused-all graph $P $N"
python -m gt4py_dwarf_p_cloudsc2_tl_ad_amd.drivers.run_symmetry_test --num-cols $N --num-runs 3 --precision $P --fused --graph 2>&1 | grep -E "^The symmetry|^The test|Traceback|Error" | tr '\n' ' '; echo " <- symmetry fused graph $P $N"
done; done
