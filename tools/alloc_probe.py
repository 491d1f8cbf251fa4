"""How much VRAM a hipMalloc of a given size really takes on this box (free memory before / after)."""
import torch
def free():
    return torch.cuda.mem_get_info(0)[0]
torch.cuda.init()
for mb in (1, 3, 11, 33, 63, 65, 66.5, 100, 129, 135, 200, 257, 513, 1000, 1025, 1500, 2049, 5000):
    n = int(mb * 1024 * 1024)
    f0 = free()
    bufs = [torch.cuda.caching_allocator_alloc(n) for _ in range(4)]
    f1 = free()
    for b in bufs:
        torch.cuda.caching_allocator_delete(b)
    torch.cuda.empty_cache()
    print(f"{mb:8.1f} MB requested -> {(f0 - f1) / 4 / 1048576:9.1f} MB taken")
