tools/profile_gpu.sh 4k
tools/profile_gpu.sh 1080p --workload 1080p
tools/profile_gpu.sh 8k-half --workload 8k-half
PROFILE_PROG=tools/bench_encode.py tools/profile_gpu.sh encode --frames-per-launch 32
