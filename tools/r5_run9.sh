for pz in 0xFF 0x00 0x7F; do
  echo "== poison $pz"; MISSLAP_DEBUG_POISON=$pz timeout -k 10 300 python -m pytest @tools/r5_shape_tests.txt -q -p no:cacheprovider 2>&1 | grep -E "FAILED|passed|failed" | head -20
done > gpurun_out/r5_t9.log 2>&1; cat gpurun_out/r5_t9.log
