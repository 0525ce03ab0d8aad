set -o pipefail
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
# usage (on the GPU box): bash tools/refresh_round.sh [skip]   -- the round's timelines, rank shares, size sweep, section shares and the table of all workloads, into gpurun_out/
if [ "$1" != skip ]; then
(echo "# tools/wave_timeline.py 10: config 1 (1200x800x10), brute-force path then camera-ray lists; diagnostic build, with the block-shared end of a launch"; RT_AMD_PRIMARY=sweep timeout -k 10 120 python tools/wave_timeline.py 10 && timeout -k 10 120 python tools/wave_timeline.py 10) 2>&1 | grep -v amdgpu.ids > gpurun_out/r03_wave_timeline.txt || exit 1
(echo "# tools/rank_share.py 8, RT_AMD_PRIMARY=sweep (brute force), then with camera-ray lists"; RT_AMD_PRIMARY=sweep timeout -k 10 200 python tools/rank_share.py 8 && timeout -k 10 200 python tools/rank_share.py 8) 2>&1 | grep -v amdgpu.ids > gpurun_out/r03_rank_share.txt || exit 1
fi
(echo "# tools/size_sweep.py, RT_AMD_PRIMARY=sweep: time of all kernels of a render call for the whole 1200x800 final-scene image against samples per pixel"; RT_AMD_PRIMARY=sweep timeout -k 10 200 python tools/size_sweep.py) 2>&1 | grep -v amdgpu.ids > gpurun_out/r03_size_sweep.txt || exit 1
(echo "# tools/section_profile.py 50 with RT_AMD_PRIMARY=sweep (every ray swept; camera rays from k_primary), diagnostic build, lower occupancy"; RT_AMD_PRIMARY=sweep timeout -k 10 200 python tools/section_profile.py 50) 2>&1 | grep -v amdgpu.ids > gpurun_out/r03_section_profile_new.txt || exit 1
bash tools/final_table.sh > gpurun_out/r03_all_workloads.txt 2>&1 || exit 1
echo refresh done
