#!/bin/bash
# the driver's bench command, timed from outside (wall seconds per run), a few times in a row; then --pmc off and --extras full
out=gpurun_out/${1:-r5a}; mkdir -p $out
for i in 1 2 3; do
  t0=$(date +%s.%N)
  python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/line$i.json 2> $out/err$i.txt
  echo "run $i rc=$? wall $(echo "$(date +%s.%N) - $t0" | bc) s, line $(wc -c < $out/line$i.json) bytes"
  cp gpurun_out/bench_extras.json $out/extras$i.json
done
t0=$(date +%s.%N)
python3 bench.py --gpus 1 --steps 20 --warmup 5 --pmc off > $out/line_nopmc.json 2> $out/err_nopmc.txt
echo "pmc off rc=$? wall $(echo "$(date +%s.%N) - $t0" | bc) s"
t0=$(date +%s.%N)
python3 bench.py --extras full --extras-file $out/full.json > $out/line_full.json 2> $out/err_full.txt
echo "extras full rc=$? wall $(echo "$(date +%s.%N) - $t0" | bc) s, line $(wc -c < $out/line_full.json) bytes"
tail -3 $out/err_full.txt
cat $out/line1.json
