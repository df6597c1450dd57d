#!/bin/bash
# C3 headline step: round 4's library, bd015aa (before the store / mask / block-size changes) and HEAD, order drawn per round, 8 rounds
bash profiles/tools/ab_libs.sh gpurun_out/exp22_c3_bisect.log 8 "r04=build/variants/lib_r04.so bd015aa=build/variants/lib_bd015aa.so head=HEAD" "c3"
