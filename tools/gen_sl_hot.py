#!/usr/bin/env python3
"""Generates eorb_slam_amd/csrc/sl_hot_asm.h: the body of sl_hot_kernel (ev_slots.hip) as one inline-asm string plus its clobber list.

The kernel keeps a tile position's rows IN REGISTERS (row r = VGPR r, lane = pixel; 240 rows + a zero row) and walks a long
(slice, tile) list with the VGPR index mode of gfx9-class ISAs.  An entry is 16 bits, 0x1000 | slot: exactly the low half of M0 in
that mode (M0[7:0] = index, M0[15:12] = which operands are relative: SRC0), so an entry costs ONE scalar instruction (s_mov_b32 m0 /
s_lshr_b32 m0, .., 16 -- M0[31:16] is not looked at) and one v_add_f32 acc, v[0 + M0], acc: no LDS read, no address arithmetic, no
byte extraction (the one-byte entries of round 3 cost 1.75 scalar instructions each, and the scalar ALU, one per CU, was the binding
pipe).

Entries come through scalar loads.  Scalar loads return out of order, so `s_waitcnt lgkmcnt(0)` is their only safe wait and a load can
be in flight for one phase at most: two buffers of 32 SGPRs (64 entries, ~550 cycles of adds) -- every SGPR the wave has goes to them,
the item's state sits in lanes of v253 meanwhile.  That hides an L2 hit, not HBM: one vector load per two phases touches the lines
PF_AHEAD bytes further on (its result is never read), so the scalar loads find them in the L2.
python3 tools/gen_sl_hot.py > eorb_slam_amd/csrc/sl_hot_asm.h
"""
NROWS = 240            # rows held in v0..v239; v240 = 0.0 (the null row); v241 acc; v242 4*lane; v243 px; v244 py; v245.. temps
PF_AHEAD = 3968        # bytes the L2 prefetch runs ahead of the scalar loads (immediate offset: < 4096)
DW_BYTES = 16          # code bytes per entry dword (the tail jumps into the sequence)
BUF_A, BUF_B = 36, 68  # s[36:67], s[68:99] (tuples of 16 must start at a multiple of 4)
# scalars: s33 temp, s34 byte offset of the next request, s35 phases left, s[100:101] the list
# between items the buffers are free: s[36:51] bucket counts, s[52:59] the descriptor, s60 ticket, s61 bucket, s[62:63] exec, s[64:65] addresses
L = []
def a(s): L.append(s)

def process(buf0):
    """64 entries held in s[buf0 .. buf0+31]"""
    for d in range(32):
        s = "s%d" % (buf0 + d)
        a("s_mov_b32 m0, %s" % s); a("v_add_f32 v241, v0, v241")
        a("s_lshr_b32 m0, %s, 16" % s); a("v_add_f32 v241, v0, v241")

def load_buf(buf0, first=False):
    if first:
        a("s_load_dwordx16 s[%d:%d], s[100:101], 0x0" % (buf0, buf0 + 15)); a("s_load_dwordx16 s[%d:%d], s[100:101], 0x40" % (buf0 + 16, buf0 + 31))
    else:
        a("s_load_dwordx16 s[%d:%d], s[100:101], s34" % (buf0, buf0 + 15)); a("s_add_u32 s33, s34, 0x40")
        a("s_load_dwordx16 s[%d:%d], s[100:101], s33" % (buf0 + 16, buf0 + 31)); a("s_add_u32 s34, s34, 0x80")

# ---- prologue ----
a("s_setprio 3")                                                           # a long list is a serial chain: its wave goes first at the issue arbiter
a("v_mbcnt_lo_u32_b32 v242, -1, 0"); a("v_mbcnt_hi_u32_b32 v242, -1, v242")
a("v_and_b32 v249, 1, v242"); a("v_lshlrev_b32 v249, 7, v249")           # prefetch pattern of the loop: two 128-byte lines
a("v_lshlrev_b32 v252, 6, v242")                                           # ... and of a list's start: 64 x 64 bytes
a("v_and_b32 v243, 7, v242"); a("v_lshrrev_b32 v244, 3, v242"); a("v_lshlrev_b32 v242, 2, v242")
a("v_mov_b32 v246, v243"); a("v_mov_b32 v247, v244")                      # lx, ly
a("s_load_dwordx16 s[36:51], %[hcnt], 0x0")                                # items per bucket (16 buckets, heaviest first)
a("s_waitcnt lgkmcnt(0)")
# next ticket: lane 0 takes one (global_atomic_add with return), broadcast
a("SLH_ITEM:")
a("v_mov_b32 v245, 1"); a("v_mov_b32 v248, 0")
a("s_mov_b64 s[62:63], exec"); a("s_mov_b64 exec, 1")
a("global_atomic_add v245, v248, v245, %[ticket] sc0")
a("s_waitcnt vmcnt(0)")
a("s_mov_b64 exec, s[62:63]")
a("s_nop 0")
a("v_readfirstlane_b32 s60, v245")                                         # ticket
# ticket -> (bucket, index): walk the 16 counts
a("s_mov_b32 s61, 0")                                                      # bucket
for b in range(16):
    a("s_cmp_lt_u32 s60, s%d" % (36 + b)); a("s_cbranch_scc1 SLH_FOUND")
    a("s_sub_u32 s60, s60, s%d" % (36 + b)); a("s_add_u32 s61, s61, 1")
a("s_branch SLH_DONE")
a("SLH_FOUND:")
a("s_mul_i32 s61, s61, %[hcap]"); a("s_add_u32 s61, s61, s60"); a("s_lshl_b32 s61, s61, 5")       # byte offset of the 32-byte descriptor
a("s_load_dwordx8 s[52:59], %[items], s61")                                # slice, tile, cnt, off_lo, off_hi, tx0, ty0, rows byte offset
a("s_waitcnt lgkmcnt(0)")
a("s_add_u32 s64, %[rows_lo], s59"); a("s_addc_u32 s65, %[rows_hi], 0")
for blk in range(NROWS // 16):
    for r in range(16):
        a("global_load_dword v%d, v242, s[64:65] offset:%d" % (blk * 16 + r, r * 256))
    a("s_add_u32 s64, s64, 0x1000"); a("s_addc_u32 s65, s65, 0")
a("v_mov_b32 v240, 0"); a("v_mov_b32 v241, 0")
a("s_add_u32 s100, %[ent_lo], s55"); a("s_addc_u32 s101, %[ent_hi], s56")   # the list
a("global_load_dword v250, v252, s[100:101]")                                # the list's first 4 KB towards the L2 (v250 is never read)
a("v_writelane_b32 v253, s52, 0"); a("v_writelane_b32 v253, s54, 1")      # the item's slice, entries, tile origin: kept in v253
a("v_writelane_b32 v253, s57, 2"); a("v_writelane_b32 v253, s58, 3")
a("s_lshr_b32 s35, s54, 6")                                                # full 64-entry phases
load_buf(BUF_A, first=True)
a("s_mov_b32 s34, 0x80")                                                   # byte offset of the next block to request
a("s_mov_b32 s33, 0")
a("s_waitcnt vmcnt(0)")
a("s_set_gpr_idx_on s33, gpr_idx(SRC0)")
a("SLH_LOOP:")
a("v_add_u32 v251, s34, v249"); a("global_load_dword v250, v251, s[100:101] offset:%d" % PF_AHEAD)      # the 256 bytes PF_AHEAD past the next request
for mine, other in ((BUF_A, BUF_B), (BUF_B, BUF_A)):
    a("s_sub_u32 s35, s35, 1"); a("s_cbranch_scc1 SLH_TAIL")              # (borrow: no full phase left)
    a("s_waitcnt lgkmcnt(0)")
    load_buf(other)
    process(mine)
a("s_branch SLH_LOOP")
# ---- tail: the last (cnt mod 64) entries = the last halves of the 128-byte block that ends at the list's (4-byte rounded) end ----
a("SLH_TAIL:")
a("s_waitcnt lgkmcnt(0)")
a("s_set_gpr_idx_off")
a("v_readlane_b32 s33, v253, 1")                                           # entries of the list
a("s_and_b32 s35, s33, 63"); a("s_cmp_eq_u32 s35, 0"); a("s_cbranch_scc1 SLH_STORE")
a("s_add_u32 s34, s33, 1"); a("s_and_b32 s34, s34, -2"); a("s_lshl_b32 s34, s34, 1")     # bytes of the list, rounded up to a dword
a("s_add_u32 s100, s100, s34"); a("s_addc_u32 s101, s101, 0"); a("s_sub_u32 s100, s100, 0x80"); a("s_subb_u32 s101, s101, 0")
load_buf(BUF_A, first=True)
a("s_waitcnt lgkmcnt(0)")
# an odd count: the upper half of the last dword is past the end of the list -> the null row (0x1000 | 240)
a("s_bitcmp1_b32 s33, 0"); a("s_cbranch_scc0 SLH_TAILGO")
a("s_and_b32 s%d, s%d, 0xffff" % (BUF_A + 31, BUF_A + 31)); a("s_or_b32 s%d, s%d, 0x10f00000" % (BUF_A + 31, BUF_A + 31))
a("SLH_TAILGO:")
# jump to dword (32 - ndw) of the sequence below, ndw = dwords that hold tail entries
a("s_add_u32 s33, s35, 1"); a("s_lshr_b32 s33, s33, 1")                   # ndw in 1..32
a("s_sub_u32 s33, 32, s33"); a("s_mul_i32 s33, s33, %d" % DW_BYTES)
a("s_mov_b32 s34, 0")
a("s_set_gpr_idx_on s34, gpr_idx(SRC0)")
a("s_getpc_b64 s[34:35]")
a("s_add_u32 s34, s34, s33"); a("s_addc_u32 s35, s35, 0")
a("s_add_u32 s34, s34, 20"); a("s_addc_u32 s35, s35, 0")                   # the five 4-byte instructions between s_getpc's return value and the sequence
a("s_setpc_b64 s[34:35]")
process(BUF_A)
# ---- the item's pixels ----
a("SLH_STORE:")
a("s_set_gpr_idx_off")
a("v_readlane_b32 s52, v253, 0"); a("v_readlane_b32 s57, v253, 2"); a("v_readlane_b32 s58, v253, 3")
a("v_add_u32 v243, s57, v246"); a("v_add_u32 v244, s58, v247")            # px, py
a("v_cmp_gt_u32 vcc, %[W], v243"); a("v_cmp_gt_u32 s[34:35], %[H], v244"); a("s_and_b64 s[34:35], s[34:35], vcc")
a("v_mul_lo_u32 v245, v244, %[W]"); a("v_add_u32 v245, v245, v243"); a("v_lshlrev_b32 v245, 2, v245")
a("s_mul_i32 s33, %[W], %[H]"); a("s_mul_i32 s33, s33, s52"); a("s_mul_hi_u32 s60, s33, 4"); a("s_lshl_b32 s33, s33, 2")
a("s_add_u32 s64, %[img_lo], s33"); a("s_addc_u32 s65, %[img_hi], s60")
a("s_mov_b64 s[62:63], exec"); a("s_and_b64 exec, exec, s[34:35]")
a("global_store_dword v245, v241, s[64:65]")
a("s_mov_b64 exec, s[62:63]")
# running maximum (every increment is >= 0: the largest final value): wave reduction, one atomic
a("v_mov_b32 v248, v241"); a("s_nop 1")
for sh in (1, 2, 4, 8):
    a("v_max_f32_dpp v248, v248, v248 row_shr:%d row_mask:0xf bank_mask:0xf bound_ctrl:0" % sh); a("s_nop 1")
a("v_max_f32_dpp v248, v248, v248 row_bcast:15 row_mask:0xa bank_mask:0xf"); a("s_nop 1")
a("v_max_f32_dpp v248, v248, v248 row_bcast:31 row_mask:0xc bank_mask:0xf"); a("s_nop 1")
a("v_readlane_b32 s33, v248, 63")
a("s_or_b32 s33, s33, 0x80000000")                                          # enc_f32 of a non-negative float
a("s_lshl_b32 s34, s52, 3"); a("s_add_u32 s34, s34, 4")
a("v_mov_b32 v245, s34"); a("v_mov_b32 v248, s33")
a("s_mov_b64 s[62:63], exec"); a("s_mov_b64 exec, 1")
a("global_atomic_umax v245, v248, %[mm]")
a("s_mov_b64 exec, s[62:63]")
a("s_load_dwordx16 s[36:51], %[hcnt], 0x0")                                # (the buffers held entries: the bucket counts again)
a("s_waitcnt vmcnt(0) lgkmcnt(0)")
a("s_branch SLH_ITEM")
a("SLH_DONE:")

print("// generated by tools/gen_sl_hot.py -- do not edit")
print("#define SL_HOT_NROWS %d" % NROWS)
print("#define SL_HOT_ASM \\")
for s in L:
    print('    "%s\\n" \\' % s)
print('    ""')
clob = ["v%d" % i for i in range(256)] + ["s%d" % i for i in range(33, 102)] + ["vcc", "scc", "m0", "memory"]
print("#define SL_HOT_CLOBBERS " + ", ".join('"%s"' % c for c in clob))
