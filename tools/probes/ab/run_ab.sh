timeout 100 tools/x2v_check attn | tail -4
for rep in 1 2; do for v in 12 13; do timeout 60 tools/x2v_check pattn $v 75600 5 6 | head -1; done; done
for v in 12 13; do timeout 60 tools/x2v_check pattn $v 20280 12 10 | head -1; done
for v in 12 13; do timeout 100 tools/x2v_check pattn $v 75600 40 3 | head -1; done
