# shader clock / power while one kernel loops (is the matrix pipe clock- or power-limited under real data?)
#   tools/clock_watch.sh "<conv_bench geometry>" <kind> [GHM_ABLATE value]
G="$1"; K="$2"; AB="${3:-0}"
( GHM_ABLATE=$AB python tools/conv_bench.py $G --kinds $K --reps 4000 > /tmp/cb.txt 2>&1 ) &
PID=$!
sleep 2.5
for i in 1 2 3 4 5 6; do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power|power" | tr '\n' ' '; echo
  sleep 0.3
done
wait $PID
cat /tmp/cb.txt
