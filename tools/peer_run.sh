W=$1; P=$((20000 + RANDOM % 20000))
for r in $(seq 0 $((W-1))); do python tests/peer_worker.py $r $W $P 4096 & done; wait
