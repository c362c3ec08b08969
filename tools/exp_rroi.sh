# RRoIAlign: channels per block (XCD-aware block order)
for g in 1 2 3 4 5 8; do echo -n "gpb=$g : "; OBB_RROI_GPB=$g timeout 100 python tools/bench_rroi.py; done
