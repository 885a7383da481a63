for s in 1 2 3 4 6; do python3 bench.py --steps 40 --warmup 5 --no-cpu-baseline --streams $s 2>/dev/null | tail -1 | python3 -c "
import sys,json; j=json.loads(sys.stdin.read()); t=j['throughput_mode']; print('streams',t['streams_per_gpu'],'value',t['value'],'scans/s',t['scans_per_s'])"; done
