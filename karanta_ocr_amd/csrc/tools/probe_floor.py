import ctypes as C, torch, sys
sys.path.insert(0, '/root/repo')
from karanta_ocr_amd._lib import lib
L = lib()
us = C.c_float()
torch.zeros(1, device='cuda')
st = torch.cuda.Stream()
for name, s in (("torch stream", st.cuda_stream), ("null stream?", None)):
    if s is None: continue
    for blocks in (1, 256):
        for dirty in (0, 1, 2):
            L.kr_probe_launch_floor(s, 200, blocks, dirty, C.byref(us))
            print(f"{name}: blocks={blocks} dirty={dirty}: {us.value:.2f} us/kernel")
