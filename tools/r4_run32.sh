mkdir -p gpurun_out/r4I
for t in 1300 1700 2100 2611 3200; do echo "p=6 DN_TINY_LEN=$t"; DN_TINY_LEN=$t timeout -k 10 120 python3 tools/p_sweep.py 4000 6 | tail -1; done > gpurun_out/r4I/tiny_p6.txt 2>&1
for t in 1500 2100 2700 3286 4200; do echo "p=4 DN_TINY_LEN=$t"; DN_TINY_LEN=$t timeout -k 10 120 python3 tools/p_sweep.py 4000 4 | tail -1; done > gpurun_out/r4I/tiny_p4.txt 2>&1
for t in 1100 1500 1853 2300; do echo "p=10 DN_TINY_LEN=$t"; DN_TINY_LEN=$t timeout -k 10 120 python3 tools/p_sweep.py 4000 10 | tail -1; done > gpurun_out/r4I/tiny_p10.txt 2>&1
cat gpurun_out/r4I/tiny_p6.txt gpurun_out/r4I/tiny_p4.txt gpurun_out/r4I/tiny_p10.txt | cut -c1-200
