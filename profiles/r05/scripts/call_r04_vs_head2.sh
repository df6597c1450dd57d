#!/bin/bash
# round 4's library (ae4e7e0) against HEAD's, the order of the two drawn per workload and round (profiles/tools/ab_libs.sh)
bash profiles/tools/ab_libs.sh gpurun_out/exp21_round4_vs_head_random_order.log 4 "r04=build/variants/lib_r04.so head=HEAD" "c3 c3x8 c4 c4vg c4:8 c5 c2"
