# usage: bash tools/exp_ub.sh <tag> <lib>   -- run a microbenchmark built as a shared library (extern "C" ubench_main) through ctypes
set -x
mkdir -p gpurun_out/$1
python3 -c "
import ctypes
l=ctypes.CDLL('tools/ubench/$2'); l.ubench_main()
" > gpurun_out/$1/$2.txt 2>&1
echo rc=$?
cat gpurun_out/$1/$2.txt | cut -c1-300 | tail -12
