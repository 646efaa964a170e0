"""Fork / join per env-step between two handles, captured in ONE hipGraph (no host cost): does splitting a 4097..8192-env batch into
two launches help when the halves must meet after every step?  (GPU box: python tools/forkjoin_graph_ab.py)"""
import os, sys, torch, time
sys.path.insert(0, ".")
src = open("tools/split_launch_ab.py").read().split("for n in (5120, 6144, 8192):")[0]
exec(src)
def run(parts, mp, G=16, reps=120):
    sims = []; base = 0
    for p in parts:
        sims.append(sim_of(p, mp, base=base)); base += p
    main = torch.cuda.Stream(dev)
    sides = [torch.cuda.Stream(dev) for _ in sims[1:]]
    fns = []
    for s in sims:
        acts = [torch.rand((s.n, 12), device=dev) * 2 - 1 for _ in range(8)]
        out = [torch.empty((s.n, 35), device=dev) for _ in range(2)]
        fns.append((s, acts, out))
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=main):
        cur = torch.cuda.current_stream(dev)
        for k in range(G):
            for st in sides:
                st.wait_stream(cur)                      # fork
            s0, a0, o0 = fns[0]
            s0.step_device_packed(a0[k & 7], o0[k & 1], stream=cur)
            for (s, a, o), st in zip(fns[1:], sides):
                s.step_device_packed(a[k & 7], o[k & 1], stream=st)
            for st in sides:
                cur.wait_stream(st)                      # join
    with torch.cuda.stream(main):
        for _ in range(10):
            graph.replay()
        main.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            graph.replay()
        main.synchronize()
        dt = time.perf_counter() - t0
    for s in sims: s.close()
    return dt / (reps * G) * 1e6
for parts in ((4096, 1024), (4096, 4096), (5120,), (4096,)):
    print(parts, "link, fork/join per step inside one hipGraph:", round(run(parts, _abi.MAP_LINK), 2), "us per step", flush=True)
