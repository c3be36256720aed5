#!/bin/bash
# Runs on the GPU box: the three profiled configurations of a round (tools/profile.sh each), then the summaries.
#   bash tools/profile_all.sh r02
R=${1:-r03}
bash tools/profile.sh ${R} > gpurun_out/prof_${R}.log 2>&1
bash tools/profile.sh ${R}_bf16 --dtype bf16 --batch 256 > gpurun_out/prof_${R}_bf16.log 2>&1
bash tools/profile.sh ${R}_point --variant point > gpurun_out/prof_${R}_point.log 2>&1
bash tools/profile.sh ${R}_bf16_u8 --dtype bf16 --batch 256 --renders u8 > gpurun_out/prof_${R}_bf16_u8.log 2>&1   # 8-bit renders: the stem's fetch bytes
cd /root/repo
python3 tools/summarize_profile.py gpurun_out/prof_${R} ${R} conv_glds,conv_finish_kernel,wino_kernel,wino_dual_kernel,dwino3_kernel,wino_finish_kernel,wino_input_kernel,wino_diff,wino2_input_kernel,wino2p_input_kernel,wino2_finish_flat_kernel,wino2s_finish_kernel,wino2p_finish_kernel
python3 tools/summarize_profile.py gpurun_out/prof_${R}_bf16 ${R}_bf16 conv_bf16
python3 tools/summarize_profile.py gpurun_out/prof_${R}_point ${R}_point conv_glds,conv_finish_kernel,wino_kernel,wino_dual_kernel,dwino3_kernel,wino_finish_kernel,wino_input_kernel,wino_diff,wino2_input_kernel,wino2p_input_kernel,wino2_finish_flat_kernel,wino2s_finish_kernel,wino2p_finish_kernel
python3 tools/summarize_profile.py gpurun_out/prof_${R}_bf16_u8 ${R}_bf16_u8 conv_bf16
mkdir -p gpurun_out/profiles_${R}
cp profiles/${R}_* profiles/traffic_${R}* gpurun_out/profiles_${R}/ 2>/dev/null
# the bench lines of the same snapshot (the PMC fields now match this tree's hash)
python3 bench.py > gpurun_out/profiles_${R}/${R}_bench.json 2> gpurun_out/profiles_${R}/${R}_bench_stderr.txt
ls -la gpurun_out/profiles_${R}
