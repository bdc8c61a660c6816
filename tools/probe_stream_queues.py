import torch, sys
torch.cuda.init()
x = torch.zeros(1024, device="cuda")
order = sys.argv[1] if len(sys.argv) > 1 else "create_then_use"
normal = [torch.cuda.Stream() for _ in range(8)]
high = [torch.cuda.Stream(priority=-1) for _ in range(4)]
def touch(st, k):
    with torch.cuda.stream(st):
        t = torch.zeros(1000 + k, device="cuda"); t.add_(1.0)
    torch.cuda.synchronize()
if order == "create_then_use":
    for k, st in enumerate(normal): touch(st, k)
    for k, st in enumerate(high): touch(st, 100 + k)
elif order == "reverse_use":
    for k, st in reversed(list(enumerate(normal))): touch(st, k)
    for k, st in enumerate(high): touch(st, 100 + k)
elif order == "high_first":
    for k, st in enumerate(high): touch(st, 100 + k)
    for k, st in enumerate(normal): touch(st, k)
print("done")
