set -e
for cs in 1 2 4; do timeout -k 10 60 tools/conv_bench 40 5 5 1 1 $cs; timeout -k 10 60 tools/conv_bench 40 5 5 0 1 $cs; done
for B in 17 32 64 128 177 256 355 512 1024; do for cs in 1 2 4; do echo -n "cs=$cs "; timeout -k 10 60 tools/conv_bench $B 50 5 1 0 $cs | tail -1; done; done
