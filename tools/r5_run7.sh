for k in 1 2 3; do timeout -k 10 300 python -m pytest @tools/r5_shape_tests.txt -q -p no:cacheprovider 2>&1 | tail -4; done > gpurun_out/r5_t7.log 2>&1; cat gpurun_out/r5_t7.log
