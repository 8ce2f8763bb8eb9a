N=${1:-20}
for i in $(seq 1 $N); do for r in 1 2 3 4 5 6 7 8; do python tools/loud_detect.py 2>/dev/null | grep LOUD & done; wait; done | grep -c BAD
