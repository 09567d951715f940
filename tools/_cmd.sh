timeout -k 10 200 python tools/probe_lookahead.py flow 2>&1 | grep -E "SMODE 5|0 waves x|^plain|rror|differences|rows per wave 0" | head -8
timeout -k 10 200 python tools/probe_stamps_flow.py 2>&1 | tail -9
timeout -k 10 300 python tools/ab_two_libs.py tools/libmbb_head.so mbb_emcee_amd/libmbb_hip.so 7 2>&1 | head -3
