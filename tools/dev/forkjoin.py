import os, sys, torch
sys.path.insert(0, ".")
src = open("tools/split_launch_ab.py").read().split("for n in (5120, 6144, 8192):")[0]
exec(src)
main = torch.cuda.Stream(dev)
def run(parts, mp, steps=1500):
    sims = []; base = 0
    for p in parts:
        sims.append(sim_of(p, mp, base=base)); base += p
    sts = [torch.cuda.Stream(dev) for _ in sims]
    fns = []
    for s, st in zip(sims, sts):
        acts = [torch.rand((s.n, 12), device=dev) * 2 - 1 for _ in range(8)]
        out = [torch.empty((s.n, 35), device=dev) for _ in range(2)]
        fns.append(s.bind_step_packed(acts, out, stream=st))
    fork = [torch.cuda.Event() for _ in range(2)]
    joins = [[torch.cuda.Event() for _ in sims] for _ in range(2)]
    res = None
    for count in (200, steps):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(main)
        for k in range(count):
            b = k & 1
            fork[b].record(main)
            for fn, st, j in zip(fns, sts, joins[b]):
                st.wait_event(fork[b])
                fn(k & 7, b)
                j.record(st)
                main.wait_event(j)
        e1.record(main)
        torch.cuda.synchronize()
        res = e0.elapsed_time(e1) / count * 1e3
    for s in sims: s.close()
    return res
for parts, mp in (((4096, 1024), _abi.MAP_LINK), ((4096, 4096), _abi.MAP_LINK), ((2560, 2560), _abi.MAP_LINK), ((4096,), _abi.MAP_LINK), ((16384, 16384, 16384), _abi.MAP_QUAD)):
    print(parts, NAME[mp], "fork/join per env-step:", round(run(parts, mp), 2), "us", flush=True)
