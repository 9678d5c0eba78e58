for nst in 2 3; do RDPN6D_H2_NST=$nst python tools/bench_conv_h2.py; done
RDPN6D_H2_NST=3 RDPN6D_H2_TILE=64,128 python tools/bench_conv_h2.py
RDPN6D_H2_NST=3 RDPN6D_H2_TILE=64,64 python tools/bench_conv_h2.py
RDPN6D_H2_NST=3 RDPN6D_H2_TILE=128,64 python tools/bench_conv_h2.py
python -m pytest tests/test_gpu_h2.py -q 2>&1 | tail -2
