mkdir -p gpurun_out/r4l
bash tools/sweep_classes.sh r4l "3000 3400 3700 4300" "1300 1500 1700 2100" > /dev/null 2>&1; cat gpurun_out/r4l/classes.log
python3 tools/p_sweep.py 4000 2 3 4 5 6 7 8 9 10 11 12 13 16 > gpurun_out/r4l/p_sweep.txt 2>&1; cat gpurun_out/r4l/p_sweep.txt | cut -c1-110
