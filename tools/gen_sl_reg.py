#!/usr/bin/env python3
"""Generates eorb_slam_amd/csrc/sl_reg_asm.h: the body of sl_reg_kernel (ev_slots.hip) as one inline-asm string plus its clobber list.

One wavefront = one task = a tile position: its rows are loaded INTO REGISTERS once (row r = VGPR r, lane = pixel; 240 rows + a zero
row), then the wave takes the position's (slice, tile) lists by ticket, longest first.  Entries are 16 bits wide here: slot | 0x1000,
i.e. exactly the M0 image the VGPR index mode wants (M0[7:0] = index, M0[12] = "source 0 is relative"), so ONE scalar instruction per
entry (s_mov_b32 m0 / s_lshr_b32 m0, .., 16) selects the row and one v_add_f32 acc, v[0 + M0], acc adds it -- no LDS read, no address
arithmetic, 10.6 cycles per entry for a lone wave and 4.6 ns per entry and wave with two waves per SIMD (tools/mb/gpr_idx.hip; the
8-bit form with its shift + s_set_gpr_idx_idx per entry: 13.4 cycles / 7.5 ns).  Entries come through scalar loads, 32 per
s_load_dwordx16, three buffers in rotation.  The next item's ticket is requested six phases before the end of a list, its descriptor
three phases before the end.
    python3 tools/gen_sl_reg.py > eorb_slam_amd/csrc/sl_reg_asm.h
"""
NROWS = 240            # rows held in v0..v239; v240 = 0.0 (the null row); v241 acc; v242 4*lane; v243 px; v244 py; v245.. temps
NULL_ENTRY = 0x1000 | NROWS
BYTES_PER_DWORD = 16   # code bytes of one entry dword in process() (the tail jumps into the sequence)
L = []
def a(s): L.append(s)

def process(buf0):
    """32 entries held in s[buf0 .. buf0+15]: fixed 16 bytes of code per dword"""
    for d in range(16):
        s = "s%d" % (buf0 + d)
        a("s_mov_b32 m0, %s" % s); a("v_add_f32 v241, v0, v241")
        a("s_lshr_b32 m0, %s, 16" % s); a("v_add_f32 v241, v0, v241")

def request_ticket():
    """lane 0 takes the position's next ticket (no wait); index mode must be off"""
    a("v_mov_b32 v245, 1"); a("v_mov_b32 v248, 0")
    a("s_mov_b64 s[96:97], exec"); a("s_mov_b64 exec, 1")
    a("global_atomic_add v245, v248, v245, %[ctr] sc0")
    a("s_mov_b64 exec, s[96:97]")
    a("s_mov_b32 s99, 1")                                                   # stage: ticket requested

uid = [0]
def request_desc():
    """ticket -> item index s98 (clamped to nit), its descriptor into s[40:43] (no wait); index mode must be off"""
    uid[0] += 1
    a("s_waitcnt vmcnt(0)")
    a("v_readfirstlane_b32 s98, v245")
    a("s_min_u32 s98, s98, %[nit]")
    a("s_cmp_ge_u32 s98, %[nit]"); a("s_cbranch_scc1 SLR_NODESC%d" % uid[0])
    a("s_lshl_b32 s33, s98, 4"); a("s_load_dwordx4 s[40:43], %[items], s33")
    a("SLR_NODESC%d:" % uid[0])
    a("s_mov_b32 s99, 2")                                                   # stage: descriptor requested

# ---- prologue ----
a("v_mbcnt_lo_u32_b32 v242, -1, 0"); a("v_mbcnt_hi_u32_b32 v242, -1, v242")
a("v_and_b32 v246, 7, v242"); a("v_lshrrev_b32 v247, 3, v242"); a("v_lshlrev_b32 v242, 2, v242")    # lx, ly, 4 * lane
a("v_add_u32 v243, %[tx0], v246"); a("v_add_u32 v244, %[ty0], v247")                                 # px, py
a("v_mul_lo_u32 v249, v244, %[W]"); a("v_add_u32 v249, v249, v243"); a("v_lshlrev_b32 v249, 2, v249")   # byte offset of the pixel in a slice
# the first ticket, before anything is loaded: a position that others have drained costs nothing
request_ticket()
request_desc()
a("s_waitcnt lgkmcnt(0)")
a("s_cmp_ge_u32 s98, %[nit]"); a("s_cbranch_scc1 SLR_DONE")
a("s_mov_b32 s44, %[rows_lo]"); a("s_mov_b32 s45, %[rows_hi]")
for blk in range(NROWS // 16):
    for r in range(16):
        a("global_load_dword v%d, v242, s[44:45] offset:%d" % (blk * 16 + r, r * 256))
    a("s_add_u32 s44, s44, 0x1000"); a("s_addc_u32 s45, s45, 0")
a("v_mov_b32 v240, 0")
a("s_waitcnt vmcnt(0)")
# ---- an item: descriptor in s[40:43] = { slice, entries, list byte offset (2) } ----
a("SLR_ITEM:")
a("s_mov_b32 s36, s40"); a("s_mov_b32 s37, s41"); a("s_mov_b32 s38, s42"); a("s_mov_b32 s39, s43")
a("s_and_b32 s36, s36, 0x7fffffff")
a("s_mov_b32 s99, 0")
a("v_mov_b32 v241, 0")
a("s_cmp_eq_u32 s37, 0"); a("s_cbranch_scc1 SLR_STORE")
a("s_add_u32 s44, %[ent_lo], s38"); a("s_addc_u32 s45, %[ent_hi], s39")   # the list
a("s_load_dwordx16 s[64:79], s[44:45], 0x0"); a("s_load_dwordx16 s[80:95], s[44:45], 0x40")
a("s_add_u32 s34, s44, 0x80"); a("s_addc_u32 s35, s45, 0")                 # next block to request
a("s_lshr_b32 s46, s37, 5")                                                # full 32-entry phases
a("s_cmp_ge_u32 s46, 6"); a("s_cbranch_scc1 SLR_LATER")
request_ticket()                                                            # a short list: the next ticket right away
a("SLR_LATER:")
a("s_mov_b32 s47, 0")
a("s_set_gpr_idx_on s47, gpr_idx(SRC0)")
bufs = [64, 80, 48]
a("SLR_LOOP:")
for k in range(3):
    a("s_cmp_eq_u32 s46, 0"); a("s_cbranch_scc1 SLR_TAIL")
    a("s_cmp_lg_u32 s46, 6"); a("s_cbranch_scc1 SLR_NOA%d" % k)
    a("s_set_gpr_idx_off")
    request_ticket()
    a("s_set_gpr_idx_on s47, gpr_idx(SRC0)")
    a("SLR_NOA%d:" % k)
    a("s_cmp_lg_u32 s46, 3"); a("s_cbranch_scc1 SLR_NOB%d" % k)
    a("s_cmp_lg_u32 s99, 1"); a("s_cbranch_scc1 SLR_NOB%d" % k)
    a("s_set_gpr_idx_off")
    request_desc()
    a("s_set_gpr_idx_on s47, gpr_idx(SRC0)")
    a("SLR_NOB%d:" % k)
    a("s_waitcnt lgkmcnt(0)")
    nb = bufs[(k + 2) % 3]
    a("s_load_dwordx16 s[%d:%d], s[34:35], 0x0" % (nb, nb + 15)); a("s_add_u32 s34, s34, 0x40"); a("s_addc_u32 s35, s35, 0")
    process(bufs[k])
    a("s_sub_u32 s46, s46, 1")
a("s_branch SLR_LOOP")
# ---- tail: the last (cnt mod 32) entries = the last dwords of the 64-byte block that ends at the list's (4-byte rounded) end.
#      (a list shorter than 64 bytes: the block starts before the list; the buffer carries 64 bytes of front padding) ----
a("SLR_TAIL:")
a("s_waitcnt lgkmcnt(0)")
a("s_and_b32 s46, s37, 31"); a("s_cmp_eq_u32 s46, 0"); a("s_cbranch_scc1 SLR_TAILDONE")
a("s_lshl_b32 s47, s37, 1"); a("s_add_u32 s47, s47, 3"); a("s_and_b32 s47, s47, -4")   # bytes of the list rounded up to a dword
a("s_add_u32 s34, s44, s47"); a("s_addc_u32 s35, s45, 0"); a("s_sub_u32 s34, s34, 0x40"); a("s_subb_u32 s35, s35, 0")
a("s_load_dwordx16 s[64:79], s[34:35], 0x0")
a("s_waitcnt lgkmcnt(0)")
# an odd count: the upper half of the last dword is past the end of the list -> the null row
a("s_bitcmp1_b32 s37, 0"); a("s_cbranch_scc0 SLR_TAILGO")
a("s_and_b32 s79, s79, 0xffff"); a("s_or_b32 s79, s79, 0x%x" % (NULL_ENTRY << 16))
a("SLR_TAILGO:")
# jump to dword (16 - ndw) of the sequence below, ndw = dwords that hold tail entries
a("s_and_b32 s33, s47, 63"); a("s_cmp_eq_u32 s33, 0"); a("s_cselect_b32 s33, 64, s33"); a("s_lshr_b32 s33, s33, 2")   # ndw in 1..16
a("s_sub_u32 s33, 16, s33"); a("s_lshl_b32 s33, s33, 4")                   # x BYTES_PER_DWORD
a("s_getpc_b64 s[34:35]")
a("s_add_u32 s34, s34, s33"); a("s_addc_u32 s35, s35, 0")
a("s_add_u32 s34, s34, 20"); a("s_addc_u32 s35, s35, 0")                   # the five 4-byte instructions between s_getpc's return value and the sequence
a("s_setpc_b64 s[34:35]")
process(64)
a("SLR_TAILDONE:")
a("s_set_gpr_idx_off")
# ---- the item's pixels ----
a("SLR_STORE:")
a("s_mul_i32 s33, %[W], %[H]"); a("s_mul_hi_u32 s47, s33, s36"); a("s_mul_i32 s46, s33, s36"); a("s_lshl_b64 s[46:47], s[46:47], 2")
a("s_add_u32 s44, %[img_lo], s46"); a("s_addc_u32 s45, %[img_hi], s47")
a("v_cmp_gt_u32 vcc, %[W], v243"); a("v_cmp_gt_u32 s[34:35], %[H], v244"); a("s_and_b64 s[34:35], s[34:35], vcc")     # lanes inside the image
a("s_mov_b64 s[46:47], exec"); a("s_and_b64 exec, exec, s[34:35]")
a("global_store_dword v249, v241, s[44:45]")
a("s_mov_b64 exec, s[46:47]")
a("s_cmp_eq_u32 s37, 0"); a("s_cbranch_scc1 SLR_NEXT")                     # an empty list offers nothing to the running extremes
# running maximum (every increment is >= 0: the largest final value): wave reduction, one atomic
a("v_mov_b32 v248, v241"); a("s_nop 1")
for sh in (1, 2, 4, 8):
    a("v_max_f32_dpp v248, v248, v248 row_shr:%d row_mask:0xf bank_mask:0xf bound_ctrl:0" % sh); a("s_nop 1")
a("v_max_f32_dpp v248, v248, v248 row_bcast:15 row_mask:0xa bank_mask:0xf"); a("s_nop 1")
a("v_max_f32_dpp v248, v248, v248 row_bcast:31 row_mask:0xc bank_mask:0xf"); a("s_nop 1")
a("v_readlane_b32 s33, v248, 63")
a("s_or_b32 s33, s33, 0x80000000")                                          # enc_f32 of a non-negative float
a("s_lshl_b32 s34, s36, 3"); a("s_add_u32 s34, s34, 4")
a("v_mov_b32 v250, s34"); a("v_mov_b32 v248, s33")
a("s_mov_b64 s[46:47], exec"); a("s_mov_b64 exec, 1")
a("global_atomic_umax v250, v248, %[mm]")
a("s_mov_b64 exec, s[46:47]")
# ---- the next item: whatever of ticket / descriptor the list was too short to request ----
a("SLR_NEXT:")
a("s_cmp_lg_u32 s99, 0"); a("s_cbranch_scc1 SLR_HAVET")
request_ticket()
a("SLR_HAVET:")
a("s_cmp_lg_u32 s99, 1"); a("s_cbranch_scc1 SLR_HAVED")
request_desc()
a("SLR_HAVED:")
a("s_waitcnt lgkmcnt(0)")
a("s_cmp_lt_u32 s98, %[nit]"); a("s_cbranch_scc1 SLR_ITEM")
a("SLR_DONE:")
a("s_waitcnt vmcnt(0) lgkmcnt(0)")

print("// generated by tools/gen_sl_reg.py -- do not edit")
print("#define SL_REG_NROWS %d" % NROWS)
print("#define SL_REG_ENTRY_FLAG 0x1000u")
print("#define SL_REG_ASM \\")
for s in L:
    print('    "%s\\n" \\' % s)
print('    ""')
clob = ["v%d" % i for i in range(256)] + ["s%d" % i for i in range(33, 100)] + ["vcc", "scc", "m0", "memory"]
print("#define SL_REG_CLOBBERS " + ", ".join('"%s"' % c for c in clob))
