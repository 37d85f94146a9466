import ctypes as C
import torch
hip = C.CDLL("libamdhip64.so")
hip.hipMemsetAsync.argtypes = [C.c_void_p, C.c_int, C.c_size_t, C.c_void_p]
hip.hipMemsetAsync.restype = C.c_int
s = torch.cuda.Stream()
for nbytes, val, off in ((4, 0, 0), (4, 0, 4), (8, 0, 0), (64, 0, 0), (4, 0xAB, 0), (24, 0, 0), (4, 0, 12)):
    buf = torch.full((32,), 0x09090909, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        st = torch.cuda.current_stream().cuda_stream
        rc = hip.hipMemsetAsync(buf.data_ptr() + off, val, nbytes, st)
    buf.fill_(0x07070707)
    torch.cuda.synchronize()
    g.replay()
    torch.cuda.synchronize()
    print("graph  memset %2d bytes value %#x at +%d: rc %d ->" % (nbytes, val, off, rc), [hex(v & 0xFFFFFFFF) for v in buf[:8].tolist()])
    buf.fill_(0x07070707)
    torch.cuda.synchronize()
    rc = hip.hipMemsetAsync(buf.data_ptr() + off, val, nbytes, s.cuda_stream)
    s.synchronize()
    print("direct memset %2d bytes value %#x at +%d: rc %d ->" % (nbytes, val, off, rc), [hex(v & 0xFFFFFFFF) for v in buf[:8].tolist()])
