import ctypes, sys, torch, json
sys.path.insert(0, ".")
sys.path.insert(0, "tools")
import _diag
lib = _diag.use_diag()
dev = torch.device("cuda:0")
sink = torch.zeros(4, dtype=torch.int32, device=dev)
P = lambda t: ctypes.c_void_p(t.data_ptr())
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for mb in (16, 512):
    buf = torch.randint(0, 255, (mb << 20,), dtype=torch.uint8, device=dev)
    for mode in (0, 1):
        for seg, stride in ((64, 8192), (128, 8192), (128, 8192 + 256), (256, 8192), (1024, 1024)):
            iters, blocks = 2048, 256
            def run():
                assert lib.tamd_bw_probe(P(buf), buf.numel(), seg, stride, iters, mode, blocks, P(sink), st) == 0
            for _ in range(2): run()
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record(); run(); run(); e.record(); torch.cuda.synchronize()
            t = s.elapsed_time(e) / 2 * 1e-3
            total = blocks * 8 * iters * 1024
            print(json.dumps({"buf_MB": mb, "mode": "lds-dma" if mode == 0 else "reg+ds_write", "seg": seg, "row_stride": stride,
                              "TBps": round(total / t / 1e12, 2), "GBps_per_CU": round(total / t / 1e9 / 256, 1)}), flush=True)
