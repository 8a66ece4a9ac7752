import torch, ctypes
print("torch priority_range:", torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else None)
hip = ctypes.CDLL("libamdhip64.so")
lo, hi = ctypes.c_int(), ctypes.c_int()
print(hip.hipDeviceGetStreamPriorityRange(ctypes.byref(lo), ctypes.byref(hi)), "least", lo.value, "greatest", hi.value)
for p in (-2, -1, 0, 1, 2):
    try:
        s = torch.cuda.Stream(priority=p)
        print(p, "->", s.priority)
    except Exception as e:
        print(p, "error", e)
