import os, sys, torch
sys.path.insert(0, "tools"); sys.path.insert(0, ".")
import importlib.util
spec = importlib.util.spec_from_file_location("sl", "tools/split_launch_ab.py")
src = open("tools/split_launch_ab.py").read().split("for n in (5120, 6144, 8192):")[0]
exec(src)
S = lambda: torch.cuda.Stream(dev)
for n in (5120, 8192):
    w = sim_of(n, _abi.MAP_LINK); t = time_streams([(w, S())], steps)[0]; w.close()
    print(f"{n} envs, ONE launch of the link kernel: {t:.2f} us", flush=True)
a, b = sim_of(4096, _abi.MAP_LINK), sim_of(1024, _abi.MAP_LINK, base=4096)
st = S()
ta, tb = time_streams([(a, st), (b, st)], steps)
print(f"link 4096 + link 1024 on ONE stream (serial): {ta:.2f} us per pair", flush=True)
a.close(); b.close()
# three link launches (12288 envs) on three streams; quad 2 x 16384 on two streams vs pair 32768; quad 8192 + 8192
for parts, mp in (((4096, 4096, 4096), _abi.MAP_LINK), ((16384, 16384), _abi.MAP_QUAD), ((8192, 8192), _abi.MAP_QUAD), ((16384, 16384, 16384), _abi.MAP_QUAD), ((32768, 32768), _abi.MAP_PAIR)):
    sims = []; base = 0
    for p in parts:
        sims.append(sim_of(p, mp, base=base)); base += p
    ts = time_streams([(s, S()) for s in sims], steps)
    tot = sum(parts)
    w = sim_of(tot); t1 = time_streams([(w, S())], steps)[0]; nm = NAME[w.mapping]; w.close()
    print(f"{'+'.join(map(str, parts))} {NAME[sims[0].mapping]} on {len(parts)} streams: {max(ts):.2f} us (per stream {[round(x, 2) for x in ts]}) | one AUTO launch of {tot} ({nm}): {t1:.2f} us", flush=True)
    for s in sims: s.close()
