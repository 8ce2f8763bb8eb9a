# fresh single processes of the loudness detector beside a process that keeps every CU's LDS full of 1e30 (then of 0.5)
for V in 1e30 0.5; do
[ -x tools/ubench/lds_noise ] || hipcc --offload-arch=gfx950 -O2 -o tools/ubench/lds_noise tools/ubench/lds_noise.hip 2>/dev/null
tools/ubench/lds_noise 60 $V > /dev/null &
NP=$!
sleep 1
bad=0; tot=0
for i in $(seq 1 14); do r=$(python tools/loud_detect.py 2>/dev/null | grep LOUD); tot=$((tot+1)); case "$r" in *BAD*) bad=$((bad+1)); echo "$r";; esac; done
kill $NP 2>/dev/null; wait $NP 2>/dev/null
echo "LDS noise $V: $bad bad of $tot"
done
