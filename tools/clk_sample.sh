mkdir -p gpurun_out/clk
python tools/prefill_quick.py 2048 300 > gpurun_out/clk/run.txt 2>&1 &
PID=$!
sleep 14
for i in 1 2 3 4 5 6 7 8 9 10 11 12; do
  rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -i "sclk\|mclk\|power\|junction\|fclk" | tr '\n' ';' >> gpurun_out/clk/smi.txt; echo >> gpurun_out/clk/smi.txt
  sleep 0.4
done
wait $PID
echo idle >> gpurun_out/clk/smi.txt
sleep 2
rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|mclk\|power" | tr '\n' ';' >> gpurun_out/clk/smi.txt
cat gpurun_out/clk/run.txt; cat gpurun_out/clk/smi.txt | cut -c1-700
