SRK_BFW=1 SRK_ROWSW=1 timeout 600 python tools/fuzz_conv.py 300 7 2>&1 | tail -3
SRK_BFW=1 SRK_ROWSW=1 SRK_BF3_DIRECT=0 timeout 600 python tools/fuzz_conv.py 300 11 2>&1 | tail -3
timeout 600 python tools/fuzz_conv.py 60 3 big 2>&1 | tail -3
timeout 600 python tools/fuzz_nets.py 2>&1 | tail -3
