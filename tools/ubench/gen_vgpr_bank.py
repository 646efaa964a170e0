#!/usr/bin/env python3
"""Generates tools/ubench/vgpr_bank.hip: does gfx950 charge a VALU instruction for reading several source VGPRs of one bank
(register number mod 4)?  Every kernel runs a loop of 64 independent instructions (8 destinations v40..v47 that nothing reads,
sources v8..v31 that nothing writes) whose register numbers are fixed in inline assembly; only the bank pattern differs.
Prints cycles per instruction per SIMD at 1, 2 and 4 waves per SIMD."""
PATTERNS = [
    # name, instruction template with {d} {a} {b} {c}, list of (a, b, c) source triples (8 of them)
    ("fma   a,b,c in 3 banks", "v_fma_f32 v{d}, v{a}, v{b}, v{c}", [(8 + i, 13 + i, 18 + i) for i in range(8)]),          # banks i, i+1, i+2
    ("fma   a,b same bank", "v_fma_f32 v{d}, v{a}, v{b}, v{c}", [(8 + i, 12 + i, 17 + i) for i in range(8)]),               # a%4 == b%4
    ("fma   a,c same bank", "v_fma_f32 v{d}, v{a}, v{b}, v{c}", [(8 + i, 13 + i, 16 + i) for i in range(8)]),
    ("fma   b,c same bank", "v_fma_f32 v{d}, v{a}, v{b}, v{c}", [(8 + i, 13 + i, 17 + i) for i in range(8)]),
    ("fma   a,b,c one bank", "v_fma_f32 v{d}, v{a}, v{b}, v{c}", [(8 + i, 12 + i, 16 + i) for i in range(8)]),
    ("fma   a=b (one register twice), c other", "v_fma_f32 v{d}, v{a}, v{a}, v{c}", [(8 + i, 0, 13 + i) for i in range(8)]),
    ("mul   a,b in 2 banks", "v_mul_f32_e32 v{d}, v{a}, v{b}", [(8 + i, 13 + i, 0) for i in range(8)]),
    ("mul   a,b same bank", "v_mul_f32_e32 v{d}, v{a}, v{b}", [(8 + i, 12 + i, 0) for i in range(8)]),
    ("fmac  d,a,b in 3 banks", "v_fmac_f32_e32 v{d}, v{a}, v{b}", [(9 + i, 14 + i, 0) for i in range(8)]),                 # d = 40 + i: bank i
    ("fmac  a,b same bank (d other)", "v_fmac_f32_e32 v{d}, v{a}, v{b}", [(9 + i, 13 + i, 0) for i in range(8)]),
    ("fmac  d,a same bank", "v_fmac_f32_e32 v{d}, v{a}, v{b}", [(8 + i, 13 + i, 0) for i in range(8)]),
    ("fmac  d,a,b one bank", "v_fmac_f32_e32 v{d}, v{a}, v{b}", [(8 + i, 12 + i, 0) for i in range(8)]),
    ("add_dpp quad_perm, 2 banks", "v_add_f32_dpp v{d}, v{a}, v{b} quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf", [(8 + i, 13 + i, 0) for i in range(8)]),
    ("add_dpp quad_perm, same bank", "v_add_f32_dpp v{d}, v{a}, v{b} quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf", [(8 + i, 12 + i, 0) for i in range(8)]),
    ("add_dpp row_mirror, 2 banks", "v_add_f32_dpp v{d}, v{a}, v{b} row_mirror row_mask:0xf bank_mask:0xf", [(8 + i, 13 + i, 0) for i in range(8)]),
    ("add_dpp row_newbcast, 2 banks", "v_add_f32_dpp v{d}, v{a}, v{b} row_newbcast:4 row_mask:0xf bank_mask:0xf", [(8 + i, 13 + i, 0) for i in range(8)]),
    ("mov_dpp quad_perm", "v_mov_b32_dpp v{d}, v{a} quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf", [(8 + i, 0, 0) for i in range(8)]),
    ("cndmask (VOP3, SGPR mask) 2 banks", "v_cndmask_b32_e64 v{d}, v{a}, v{b}, s[10:11]", [(8 + i, 13 + i, 0) for i in range(8)]),
    ("cndmask (VOP3, SGPR mask) same bank", "v_cndmask_b32_e64 v{d}, v{a}, v{b}, s[10:11]", [(8 + i, 12 + i, 0) for i in range(8)]),
    ("rcp", "v_rcp_f32_e32 v{d}, v{a}", [(8 + i, 0, 0) for i in range(8)]),
    ("fma   literal: fmamk", "v_fmamk_f32 v{d}, v{a}, 0x3f7fbe77, v{b}", [(8 + i, 13 + i, 0) for i in range(8)]),
    ("fma   2 VGPR + SGPR, 2 banks", "v_fma_f32 v{d}, v{a}, s12, v{b}", [(8 + i, 13 + i, 0) for i in range(8)]),
    ("fma   2 VGPR + SGPR, same bank", "v_fma_f32 v{d}, v{a}, s12, v{b}", [(8 + i, 12 + i, 0) for i in range(8)]),
    # packed: 64-bit operands occupy banks (r, r+1)
    ("pk_fma pairs in banks (0,1)(2,3)(0,1)", "v_pk_fma_f32 v[{d}:{d1}], v[{a}:{a1}], v[{b}:{b1}], v[{c}:{c1}]", [(8 + 4 * (i % 2), 14 + 4 * (i % 2), 24 + 4 * (i % 2)) for i in range(8)]),
    ("pk_fma all pairs in banks (0,1)", "v_pk_fma_f32 v[{d}:{d1}], v[{a}:{a1}], v[{b}:{b1}], v[{c}:{c1}]", [(8 + 4 * (i % 2), 16 + 4 * (i % 2), 24 + 4 * (i % 2)) for i in range(8)]),
    ("pk_fma pairs (0,1)(2,3)(2,3)", "v_pk_fma_f32 v[{d}:{d1}], v[{a}:{a1}], v[{b}:{b1}], v[{c}:{c1}]", [(8 + 4 * (i % 2), 14 + 4 * (i % 2), 26 + 4 * (i % 2)) for i in range(8)]),
    ("pk_mul pairs (0,1)(2,3)", "v_pk_mul_f32 v[{d}:{d1}], v[{a}:{a1}], v[{b}:{b1}]", [(8 + 4 * (i % 2), 14 + 4 * (i % 2), 0) for i in range(8)]),
    # EXEC masks: does a wave64 instruction with an empty half skip that half's pass through the 32-lane SIMD?
    ("fmac  EXEC = lanes 0-31", "v_fmac_f32_e32 v{d}, v{a}, v{b}", [(9 + i, 14 + i, 0) for i in range(8)], "0xffffffff"),
    ("fma   EXEC = lanes 0-31", "v_fma_f32 v{d}, v{a}, v{b}, v{c}", [(8 + i, 13 + i, 18 + i) for i in range(8)], "0xffffffff"),
    ("add_dpp quad_perm, EXEC = lanes 0-31", "v_add_f32_dpp v{d}, v{a}, v{b} quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf", [(8 + i, 13 + i, 0) for i in range(8)], "0xffffffff"),
    ("fmac  EXEC = lanes 0-15", "v_fmac_f32_e32 v{d}, v{a}, v{b}", [(9 + i, 14 + i, 0) for i in range(8)], "0xffff"),
    ("fmac  EXEC = lanes 0-15 and 32-47", "v_fmac_f32_e32 v{d}, v{a}, v{b}", [(9 + i, 14 + i, 0) for i in range(8)], "0x0000ffff0000ffff"),
    # dependent chains (dst = src0): round 2's "two interleaved chains" ran at 4.0-4.4 cycles per SIMD slot at every occupancy
    ("fma   2 chains dst=src0, shared b,c", "v_fma_f32 v{d}, v{d}, v{b}, v{c}", [(0, 8, 13)] * 8, None, [40, 41] * 4),
    ("fma   2 chains dst=src0, own b,c", "v_fma_f32 v{d}, v{d}, v{b}, v{c}", [(0, 8, 13), (0, 10, 15)] * 4, None, [40, 41] * 4),
    ("fma   4 chains dst=src0, shared b,c", "v_fma_f32 v{d}, v{d}, v{b}, v{c}", [(0, 8, 13)] * 8, None, [40, 41, 42, 43] * 2),
    ("fma   8 chains dst=src0, shared b,c", "v_fma_f32 v{d}, v{d}, v{b}, v{c}", [(0, 8, 13)] * 8, None, [40, 41, 42, 43, 44, 45, 46, 47]),
    ("fmac  2 chains, shared a,b", "v_fmac_f32_e32 v{d}, v{a}, v{b}", [(8, 13, 0)] * 8, None, [40, 41] * 4),
    ("pk_mul pairs (0,1)(0,1)", "v_pk_mul_f32 v[{d}:{d1}], v[{a}:{a1}], v[{b}:{b1}]", [(8 + 4 * (i % 2), 16 + 4 * (i % 2), 0) for i in range(8)]),
]

HEAD = r'''// GENERATED by gen_vgpr_bank.py -- VGPR bank-conflict / per-opcode issue-cost microbenchmark for gfx950 (see the generator).
// build: hipcc -O3 --offload-arch=gfx950 -o vgpr_bank vgpr_bank.hip ; run: ./vgpr_bank
#include <hip/hip_runtime.h>
#include <cstdio>
#define ITERS 4096
'''

KERNEL = r'''__global__ __launch_bounds__(64) void k%(idx)d(float *out, unsigned long long *clk) {
    unsigned long long r0 = wall_clock64();
    asm volatile(
%(init)s
%(execset)s        "s_mov_b64 s[10:11], 0x5555\n\t"
        "s_mov_b32 s12, 0x3f7fbe77\n\t"
        "s_movk_i32 s13, %(iters)d\n"
        "1:\n\t"
%(body)s
        "s_sub_u32 s13, s13, 1\n\t"
        "s_cmp_lg_u32 s13, 0\n\t"
        "s_cbranch_scc1 1b\n\t"
        "v_add_f32 v40, v40, v41\n\t"
        "global_store_dword %%0, v40, off\n\t"
        "s_waitcnt vmcnt(0)\n\t"
        "s_mov_b64 exec, -1"
        :: "v"(out + blockIdx.x * 64 + threadIdx.x)
        : %(clob)s, "s10", "s11", "s12", "s13", "scc", "memory");
    if (blockIdx.x == 0 && threadIdx.x == 0) clk[0] = wall_clock64() - r0;
}
'''


def main():
    out = [HEAD]
    names = []
    for idx, pat in enumerate(PATTERNS):
        name, tmpl, srcs = pat[:3]
        execmask = pat[3] if len(pat) > 3 else None
        dsts = pat[4] if len(pat) > 4 else None
        init = "\n".join('        "v_mov_b32 v%d, 1.0\\n\\t"' % r for r in range(8, 48))
        body = []
        for rep in range(8):
            for i in range(8):
                a, b, c = srcs[i]
                pk = "pk_" in tmpl
                d = dsts[i] if dsts else 40 + (2 * (i % 4) if pk else i)
                body.append('        "' + tmpl.format(d=d, d1=d + 1, a=a, a1=a + 1, b=b, b1=b + 1, c=c, c1=c + 1) + '\\n\\t"')
        clob = ", ".join('"v%d"' % r for r in range(8, 48))
        execset = ""
        if execmask:
            lo, hi = int(execmask, 16) & 0xffffffff, int(execmask, 16) >> 32
            execset = '        "s_mov_b32 exec_lo, 0x%x\\n\\t"\n        "s_mov_b32 exec_hi, 0x%x\\n\\t"\n' % (lo, hi)
        out.append(KERNEL % dict(idx=idx, init=init, body="\n".join(body), clob=clob, iters=4096, execset=execset))
        names.append(name)
    out.append("typedef void (*kern_t)(float *, unsigned long long *);\n")
    out.append("static kern_t KS[] = {" + ", ".join("k%d" % i for i in range(len(PATTERNS))) + "};\n")
    out.append("static const char *NAMES[] = {" + ", ".join('"%s"' % n for n in names) + "};\n")
    out.append(r'''
int main() {
    float *d; hipMalloc(&d, 8192 * 64 * 4);
    unsigned long long *clk; hipMalloc(&clk, 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int nk = sizeof(KS) / sizeof(KS[0]);
    printf("cycles per instruction at 2.4 GHz nominal (wall time of 5 launches; 1024 SIMDs); first column: per wave, second: per SIMD\n");
    for (int i = 0; i < nk; ++i) {
        for (int w = 1; w <= 4; w *= 2) {
            int blocks = 1024 * w;
            hipLaunchKernelGGL(KS[i], dim3(blocks), dim3(64), 0, 0, d, clk);
            hipEventRecord(e0);
            for (int q = 0; q < 5; q++) hipLaunchKernelGGL(KS[i], dim3(blocks), dim3(64), 0, 0, d, clk);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
            double inst = (double)ITERS * 64;
            double cyc = ms * 1e-3 * 2.4e9;
            printf("%-40s waves/SIMD %d  %7.3f ms  %6.2f per wave  %5.2f per SIMD\n", NAMES[i], w, ms, cyc / inst, cyc / inst / w);
        }
    }
    return 0;
}
''')
    open(__file__.rsplit("/", 1)[0] + "/vgpr_bank.hip", "w").write("".join(out))


if __name__ == "__main__":
    main()
