// Which gfx950 vector instructions can share a 4-cycle issue window (2 cycles each, "fast"), which take a whole window
// ("slow", 4 cycles) - and what mixed streams and register banks do to that. Companion of valu_ceiling.hip (same harness,
// wall clock only: the per-wave stamps there showed that waves of one SIMD run nearly one after the other, oldest first).
//   hipcc --offload-arch=gfx950 -O3 -o valu_classes valu_classes.hip && ./valu_classes [pmc]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float f2 __attribute__((ext_vector_type(2)));

// Every body is 256 vector instructions on hard-coded registers v[64..95] (destinations / first sources), v[96..103]
// (other sources), s[40..43] - so that register banks (index mod 4) are under control. SRC = "v96, v101" style operand text.
#define R16(F) F(64) F(65) F(66) F(67) F(68) F(69) F(70) F(71) F(72) F(73) F(74) F(75) F(76) F(77) F(78) F(79)
#define CLOB "v64","v65","v66","v67","v68","v69","v70","v71","v72","v73","v74","v75","v76","v77","v78","v79","v80","v81","v82","v83","v84","v85","v86","v87", \
             "v88","v89","v90","v91","v92","v93","v94","v95","v96","v97","v98","v99","v100","v101","v102","v103","s40","s41","s42","s43","vcc","scc"

struct kind_def { const char* name; };
enum Kind {
  K_SUB, K_MIN, K_MAX, K_AND, K_OR, K_ADD_U32, K_SUB_U32, K_LSHR_C, K_LSHL_V, K_ASHR_C, K_CVT_F32_U32, K_CVT_U32_F32, K_RNDNE, K_FLOOR, K_FFBH, K_BFE,
  K_LSHL_OR, K_LSHL_ADD, K_ADD3, K_CNDMASK_E64, K_CMP_E64, K_CMP_CNDMASK, K_ADD_SGPR, K_ADD_LIT, K_ADD_INL, K_MUL_SGPR, K_FMA_INL, K_FMA_LIT, K_FMAC, K_MOV_SGPR, K_MIN3, K_MED3,
  K_PERM, K_MUL_U24, K_MAD_U24, K_ADD_CO, K_MBCNT, K_READFIRSTLANE, K_DPP_MOV, K_SDWA_MOV, K_PKRTZ, K_XAD, K_FMA_BANK_SAME, K_FMA_BANK_DIFF, K_FMA_BANK_2SAME,
  K_PAT_FS, K_PAT_FFSS, K_PAT_FFFS, K_PAT_F8S8, K_PAT_F32S32, K_PAT_FS_OFFSET, K_COUNT };
static const char* names[K_COUNT] = {
  "v_sub_f32", "v_min_f32", "v_max_f32", "v_and_b32", "v_or_b32", "v_add_u32", "v_sub_u32", "v_lshrrev_b32 (const)", "v_lshlrev_b32 (vgpr amount)", "v_ashrrev_i32 (const)",
  "v_cvt_f32_u32", "v_cvt_u32_f32", "v_rndne_f32", "v_floor_f32", "v_ffbh_u32", "v_bfe_u32", "v_lshl_or_b32", "v_lshl_add_u32", "v_add3_u32", "v_cndmask_b32 e64 (sgpr-pair mask)",
  "v_cmp_lt_f32 e64 (sgpr-pair dst)", "v_cmp_lt_f32 vcc + v_cndmask vcc (per instr)", "v_add_f32 (sgpr src)", "v_add_f32 (literal)", "v_add_f32 (inline 1.0)", "v_mul_f32 (sgpr src)",
  "v_fma_f32 (inline 1.0)", "v_fmaak_f32 (literal addend)", "v_fmac_f32", "v_mov_b32 (sgpr src)", "v_min3_f32", "v_med3_f32", "v_perm_b32", "v_mul_u32_u24", "v_mad_u32_u24", "v_add_co_u32 (vcc)",
  "v_mbcnt_lo_u32_b32", "v_readfirstlane_b32", "v_mov_b32 dpp quad_perm", "v_mov_b32 sdwa byte", "v_cvt_pkrtz_f16_f32", "v_xad_u32", "v_fma_f32 srcs in ONE bank", "v_fma_f32 srcs in 3 banks",
  "v_fma_f32 2 of 3 srcs in one bank", "pattern F S (v_add, v_alignbit) per instr", "pattern F F S S per instr", "pattern F F F S per instr", "pattern F x8 S x8 per instr",
  "pattern F x32 S x32 per instr", "pattern F S, odd waves start with S, per instr"};

template <int KIND>
__global__ __launch_bounds__(64) void k(float* __restrict__ out, int iters, float s0, unsigned su) {
  extern __shared__ char lds_pad[];
  // initialise the fixed registers
  asm volatile("v_cvt_f32_u32 v96, %0\n v_mov_b32 v97, 0x3e800000\n v_mov_b32 v98, %0\n v_mov_b32 v99, 0x7f4a7c15\n v_mov_b32 v100, 1.0\n v_mov_b32 v101, 0.5\n v_mov_b32 v102, 3\n v_mov_b32 v103, 2.0\n"
               "s_mov_b32 s40, %1\n s_mov_b32 s41, %2\n s_mov_b32 s42, -1\n s_mov_b32 s43, 0x0000ffff\n"
               :: "v"(threadIdx.x), "s"(s0), "s"(su) : CLOB);
#define INIT(n) "v_mov_b32 v" #n ", v96\n"
  asm volatile(R16(INIT) ::: CLOB);
#undef INIT
  asm volatile("v_mov_b32 v80, v96\n v_mov_b32 v81, v97\n v_mov_b32 v82, v98\n v_mov_b32 v83, v99\n v_mov_b32 v84, v96\n v_mov_b32 v85, v97\n v_mov_b32 v86, v98\n v_mov_b32 v87, v99\n"
               "v_mov_b32 v88, v96\n v_mov_b32 v89, v97\n v_mov_b32 v90, v98\n v_mov_b32 v91, v99\n v_mov_b32 v92, v96\n v_mov_b32 v93, v97\n v_mov_b32 v94, v98\n v_mov_b32 v95, v99\n" ::: CLOB);
  const bool odd_wave = (blockIdx.x & 1u) != 0u;
  for (int it = 0; it < iters; ++it) {
#define BODY1(TEXT) asm volatile(".rept 16\n" TEXT ".endr\n" ::: CLOB)
#define ONE(op, tail) "" op " v64, v64" tail "\n" op " v65, v65" tail "\n" op " v66, v66" tail "\n" op " v67, v67" tail "\n" op " v68, v68" tail "\n" op " v69, v69" tail "\n" op " v70, v70" tail "\n" op " v71, v71" tail "\n" \
                      op " v72, v72" tail "\n" op " v73, v73" tail "\n" op " v74, v74" tail "\n" op " v75, v75" tail "\n" op " v76, v76" tail "\n" op " v77, v77" tail "\n" op " v78, v78" tail "\n" op " v79, v79" tail "\n"
    if (KIND == K_SUB) BODY1(ONE("v_sub_f32", ", v97"));
    else if (KIND == K_MIN) BODY1(ONE("v_min_f32", ", v97"));
    else if (KIND == K_MAX) BODY1(ONE("v_max_f32", ", v97"));
    else if (KIND == K_AND) BODY1(ONE("v_and_b32", ", v99"));
    else if (KIND == K_OR) BODY1(ONE("v_or_b32", ", v99"));
    else if (KIND == K_ADD_U32) BODY1(ONE("v_add_u32", ", v99"));
    else if (KIND == K_SUB_U32) BODY1(ONE("v_sub_u32", ", v99"));
    else if (KIND == K_LSHR_C) BODY1("v_lshrrev_b32 v64, 22, v64\n v_lshrrev_b32 v65, 22, v65\n v_lshrrev_b32 v66, 22, v66\n v_lshrrev_b32 v67, 22, v67\n v_lshrrev_b32 v68, 22, v68\n v_lshrrev_b32 v69, 22, v69\n v_lshrrev_b32 v70, 22, v70\n v_lshrrev_b32 v71, 22, v71\n"
                                     "v_lshrrev_b32 v72, 22, v72\n v_lshrrev_b32 v73, 22, v73\n v_lshrrev_b32 v74, 22, v74\n v_lshrrev_b32 v75, 22, v75\n v_lshrrev_b32 v76, 22, v76\n v_lshrrev_b32 v77, 22, v77\n v_lshrrev_b32 v78, 22, v78\n v_lshrrev_b32 v79, 22, v79\n");
    else if (KIND == K_LSHL_V) BODY1("v_lshlrev_b32 v64, v102, v64\n v_lshlrev_b32 v65, v102, v65\n v_lshlrev_b32 v66, v102, v66\n v_lshlrev_b32 v67, v102, v67\n v_lshlrev_b32 v68, v102, v68\n v_lshlrev_b32 v69, v102, v69\n v_lshlrev_b32 v70, v102, v70\n v_lshlrev_b32 v71, v102, v71\n"
                                     "v_lshlrev_b32 v72, v102, v72\n v_lshlrev_b32 v73, v102, v73\n v_lshlrev_b32 v74, v102, v74\n v_lshlrev_b32 v75, v102, v75\n v_lshlrev_b32 v76, v102, v76\n v_lshlrev_b32 v77, v102, v77\n v_lshlrev_b32 v78, v102, v78\n v_lshlrev_b32 v79, v102, v79\n");
    else if (KIND == K_ASHR_C) BODY1("v_ashrrev_i32 v64, 31, v64\n v_ashrrev_i32 v65, 31, v65\n v_ashrrev_i32 v66, 31, v66\n v_ashrrev_i32 v67, 31, v67\n v_ashrrev_i32 v68, 31, v68\n v_ashrrev_i32 v69, 31, v69\n v_ashrrev_i32 v70, 31, v70\n v_ashrrev_i32 v71, 31, v71\n"
                                     "v_ashrrev_i32 v72, 31, v72\n v_ashrrev_i32 v73, 31, v73\n v_ashrrev_i32 v74, 31, v74\n v_ashrrev_i32 v75, 31, v75\n v_ashrrev_i32 v76, 31, v76\n v_ashrrev_i32 v77, 31, v77\n v_ashrrev_i32 v78, 31, v78\n v_ashrrev_i32 v79, 31, v79\n");
    else if (KIND == K_CVT_F32_U32) BODY1(ONE("v_cvt_f32_u32", ""));
    else if (KIND == K_CVT_U32_F32) BODY1(ONE("v_cvt_u32_f32", ""));
    else if (KIND == K_RNDNE) BODY1(ONE("v_rndne_f32", ""));
    else if (KIND == K_FLOOR) BODY1(ONE("v_floor_f32", ""));
    else if (KIND == K_FFBH) BODY1(ONE("v_ffbh_u32", ""));
    else if (KIND == K_BFE) BODY1(ONE("v_bfe_u32", ", 3, 9"));
    else if (KIND == K_LSHL_OR) BODY1(ONE("v_lshl_or_b32", ", 4, v99"));
    else if (KIND == K_LSHL_ADD) BODY1(ONE("v_lshl_add_u32", ", 4, v99"));
    else if (KIND == K_ADD3) BODY1(ONE("v_add3_u32", ", v98, v99"));
    else if (KIND == K_CNDMASK_E64) BODY1(ONE("v_cndmask_b32", ", v97, s[42:43]"));
    else if (KIND == K_CMP_E64) BODY1("v_cmp_lt_f32 s[42:43], v64, v97\n v_cmp_lt_f32 s[42:43], v65, v97\n v_cmp_lt_f32 s[42:43], v66, v97\n v_cmp_lt_f32 s[42:43], v67, v97\n v_cmp_lt_f32 s[42:43], v68, v97\n v_cmp_lt_f32 s[42:43], v69, v97\n v_cmp_lt_f32 s[42:43], v70, v97\n v_cmp_lt_f32 s[42:43], v71, v97\n"
                                      "v_cmp_lt_f32 s[42:43], v72, v97\n v_cmp_lt_f32 s[42:43], v73, v97\n v_cmp_lt_f32 s[42:43], v74, v97\n v_cmp_lt_f32 s[42:43], v75, v97\n v_cmp_lt_f32 s[42:43], v76, v97\n v_cmp_lt_f32 s[42:43], v77, v97\n v_cmp_lt_f32 s[42:43], v78, v97\n v_cmp_lt_f32 s[42:43], v79, v97\n");
    else if (KIND == K_CMP_CNDMASK) BODY1("v_cmp_lt_f32 vcc, v64, v97\n v_cndmask_b32 v64, v64, v101, vcc\n v_cmp_lt_f32 vcc, v65, v97\n v_cndmask_b32 v65, v65, v101, vcc\n v_cmp_lt_f32 vcc, v66, v97\n v_cndmask_b32 v66, v66, v101, vcc\n v_cmp_lt_f32 vcc, v67, v97\n v_cndmask_b32 v67, v67, v101, vcc\n"
                                          "v_cmp_lt_f32 vcc, v68, v97\n v_cndmask_b32 v68, v68, v101, vcc\n v_cmp_lt_f32 vcc, v69, v97\n v_cndmask_b32 v69, v69, v101, vcc\n v_cmp_lt_f32 vcc, v70, v97\n v_cndmask_b32 v70, v70, v101, vcc\n v_cmp_lt_f32 vcc, v71, v97\n v_cndmask_b32 v71, v71, v101, vcc\n");
    else if (KIND == K_ADD_SGPR) BODY1(ONE("v_add_f32", ", s40"));
    else if (KIND == K_ADD_LIT) BODY1("v_add_f32 v64, 0x3f8ccccd, v64\n v_add_f32 v65, 0x3f8ccccd, v65\n v_add_f32 v66, 0x3f8ccccd, v66\n v_add_f32 v67, 0x3f8ccccd, v67\n v_add_f32 v68, 0x3f8ccccd, v68\n v_add_f32 v69, 0x3f8ccccd, v69\n v_add_f32 v70, 0x3f8ccccd, v70\n v_add_f32 v71, 0x3f8ccccd, v71\n"
                                      "v_add_f32 v72, 0x3f8ccccd, v72\n v_add_f32 v73, 0x3f8ccccd, v73\n v_add_f32 v74, 0x3f8ccccd, v74\n v_add_f32 v75, 0x3f8ccccd, v75\n v_add_f32 v76, 0x3f8ccccd, v76\n v_add_f32 v77, 0x3f8ccccd, v77\n v_add_f32 v78, 0x3f8ccccd, v78\n v_add_f32 v79, 0x3f8ccccd, v79\n");
    else if (KIND == K_ADD_INL) BODY1(ONE("v_add_f32", ", 1.0"));
    else if (KIND == K_MUL_SGPR) BODY1(ONE("v_mul_f32", ", s40"));
    else if (KIND == K_FMA_INL) BODY1(ONE("v_fma_f32", ", v97, 1.0"));
    else if (KIND == K_FMA_LIT) BODY1(ONE("v_fmaak_f32", ", v97, 0x3f8ccccd"));
    else if (KIND == K_FMAC) BODY1("v_fmac_f32 v64, v97, v101\n v_fmac_f32 v65, v97, v101\n v_fmac_f32 v66, v97, v101\n v_fmac_f32 v67, v97, v101\n v_fmac_f32 v68, v97, v101\n v_fmac_f32 v69, v97, v101\n v_fmac_f32 v70, v97, v101\n v_fmac_f32 v71, v97, v101\n"
                                   "v_fmac_f32 v72, v97, v101\n v_fmac_f32 v73, v97, v101\n v_fmac_f32 v74, v97, v101\n v_fmac_f32 v75, v97, v101\n v_fmac_f32 v76, v97, v101\n v_fmac_f32 v77, v97, v101\n v_fmac_f32 v78, v97, v101\n v_fmac_f32 v79, v97, v101\n");
    else if (KIND == K_MOV_SGPR) BODY1("v_mov_b32 v64, s40\n v_mov_b32 v65, s40\n v_mov_b32 v66, s40\n v_mov_b32 v67, s40\n v_mov_b32 v68, s40\n v_mov_b32 v69, s40\n v_mov_b32 v70, s40\n v_mov_b32 v71, s40\n"
                                       "v_mov_b32 v72, s40\n v_mov_b32 v73, s40\n v_mov_b32 v74, s40\n v_mov_b32 v75, s40\n v_mov_b32 v76, s40\n v_mov_b32 v77, s40\n v_mov_b32 v78, s40\n v_mov_b32 v79, s40\n");
    else if (KIND == K_MIN3) BODY1(ONE("v_min3_f32", ", v97, v103"));
    else if (KIND == K_MED3) BODY1(ONE("v_med3_f32", ", v97, v103"));
    else if (KIND == K_PERM) BODY1(ONE("v_perm_b32", ", v98, v99"));
    else if (KIND == K_MUL_U24) BODY1(ONE("v_mul_u32_u24", ", v102"));
    else if (KIND == K_MAD_U24) BODY1(ONE("v_mad_u32_u24", ", v102, v99"));
    else if (KIND == K_ADD_CO) BODY1("v_add_co_u32 v64, vcc, v64, v99\n v_add_co_u32 v65, vcc, v65, v99\n v_add_co_u32 v66, vcc, v66, v99\n v_add_co_u32 v67, vcc, v67, v99\n v_add_co_u32 v68, vcc, v68, v99\n v_add_co_u32 v69, vcc, v69, v99\n v_add_co_u32 v70, vcc, v70, v99\n v_add_co_u32 v71, vcc, v71, v99\n"
                                     "v_add_co_u32 v72, vcc, v72, v99\n v_add_co_u32 v73, vcc, v73, v99\n v_add_co_u32 v74, vcc, v74, v99\n v_add_co_u32 v75, vcc, v75, v99\n v_add_co_u32 v76, vcc, v76, v99\n v_add_co_u32 v77, vcc, v77, v99\n v_add_co_u32 v78, vcc, v78, v99\n v_add_co_u32 v79, vcc, v79, v99\n");
    else if (KIND == K_MBCNT) BODY1("v_mbcnt_lo_u32_b32 v64, s42, v64\n v_mbcnt_lo_u32_b32 v65, s42, v65\n v_mbcnt_lo_u32_b32 v66, s42, v66\n v_mbcnt_lo_u32_b32 v67, s42, v67\n v_mbcnt_lo_u32_b32 v68, s42, v68\n v_mbcnt_lo_u32_b32 v69, s42, v69\n v_mbcnt_lo_u32_b32 v70, s42, v70\n v_mbcnt_lo_u32_b32 v71, s42, v71\n"
                                    "v_mbcnt_lo_u32_b32 v72, s42, v72\n v_mbcnt_lo_u32_b32 v73, s42, v73\n v_mbcnt_lo_u32_b32 v74, s42, v74\n v_mbcnt_lo_u32_b32 v75, s42, v75\n v_mbcnt_lo_u32_b32 v76, s42, v76\n v_mbcnt_lo_u32_b32 v77, s42, v77\n v_mbcnt_lo_u32_b32 v78, s42, v78\n v_mbcnt_lo_u32_b32 v79, s42, v79\n");
    else if (KIND == K_READFIRSTLANE) BODY1("v_readfirstlane_b32 s40, v64\n v_readfirstlane_b32 s41, v65\n v_readfirstlane_b32 s40, v66\n v_readfirstlane_b32 s41, v67\n v_readfirstlane_b32 s40, v68\n v_readfirstlane_b32 s41, v69\n v_readfirstlane_b32 s40, v70\n v_readfirstlane_b32 s41, v71\n"
                                            "v_readfirstlane_b32 s40, v72\n v_readfirstlane_b32 s41, v73\n v_readfirstlane_b32 s40, v74\n v_readfirstlane_b32 s41, v75\n v_readfirstlane_b32 s40, v76\n v_readfirstlane_b32 s41, v77\n v_readfirstlane_b32 s40, v78\n v_readfirstlane_b32 s41, v79\n");
    else if (KIND == K_DPP_MOV) BODY1("v_mov_b32_dpp v64, v80 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp v65, v81 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp v66, v82 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp v67, v83 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                                      "v_mov_b32_dpp v68, v84 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp v69, v85 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp v70, v86 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp v71, v87 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                                      "v_mov_b32_dpp v72, v88 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp v73, v89 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp v74, v90 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp v75, v91 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                                      "v_mov_b32_dpp v76, v92 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp v77, v93 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp v78, v94 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp v79, v95 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n");
    else if (KIND == K_SDWA_MOV) BODY1("v_mov_b32_sdwa v64, v80 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_3\n v_mov_b32_sdwa v65, v81 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_3\n v_mov_b32_sdwa v66, v82 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_3\n v_mov_b32_sdwa v67, v83 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_3\n"
                                       "v_mov_b32_sdwa v68, v84 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_3\n v_mov_b32_sdwa v69, v85 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_3\n v_mov_b32_sdwa v70, v86 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_3\n v_mov_b32_sdwa v71, v87 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_3\n"
                                       "v_mov_b32_sdwa v72, v88 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_3\n v_mov_b32_sdwa v73, v89 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_3\n v_mov_b32_sdwa v74, v90 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_3\n v_mov_b32_sdwa v75, v91 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_3\n"
                                       "v_mov_b32_sdwa v76, v92 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_3\n v_mov_b32_sdwa v77, v93 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_3\n v_mov_b32_sdwa v78, v94 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_3\n v_mov_b32_sdwa v79, v95 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_3\n");
    else if (KIND == K_PKRTZ) BODY1(ONE("v_cvt_pkrtz_f16_f32", ", v97"));
    else if (KIND == K_XAD) BODY1(ONE("v_xad_u32", ", v98, v99"));
    // register banks: index mod 4. dst/src0 v64+4i (bank 0), other sources bank 0 (v96, v100) / banks 1, 2 (v97, v102)
    else if (KIND == K_FMA_BANK_SAME) BODY1("v_fma_f32 v64, v64, v96, v100\n v_fma_f32 v68, v68, v96, v100\n v_fma_f32 v72, v72, v96, v100\n v_fma_f32 v76, v76, v96, v100\n v_fma_f32 v80, v80, v96, v100\n v_fma_f32 v84, v84, v96, v100\n v_fma_f32 v88, v88, v96, v100\n v_fma_f32 v92, v92, v96, v100\n"
                                            "v_fma_f32 v64, v64, v96, v100\n v_fma_f32 v68, v68, v96, v100\n v_fma_f32 v72, v72, v96, v100\n v_fma_f32 v76, v76, v96, v100\n v_fma_f32 v80, v80, v96, v100\n v_fma_f32 v84, v84, v96, v100\n v_fma_f32 v88, v88, v96, v100\n v_fma_f32 v92, v92, v96, v100\n");
    else if (KIND == K_FMA_BANK_DIFF) BODY1("v_fma_f32 v64, v64, v97, v102\n v_fma_f32 v68, v68, v97, v102\n v_fma_f32 v72, v72, v97, v102\n v_fma_f32 v76, v76, v97, v102\n v_fma_f32 v80, v80, v97, v102\n v_fma_f32 v84, v84, v97, v102\n v_fma_f32 v88, v88, v97, v102\n v_fma_f32 v92, v92, v97, v102\n"
                                            "v_fma_f32 v64, v64, v97, v102\n v_fma_f32 v68, v68, v97, v102\n v_fma_f32 v72, v72, v97, v102\n v_fma_f32 v76, v76, v97, v102\n v_fma_f32 v80, v80, v97, v102\n v_fma_f32 v84, v84, v97, v102\n v_fma_f32 v88, v88, v97, v102\n v_fma_f32 v92, v92, v97, v102\n");
    else if (KIND == K_FMA_BANK_2SAME) BODY1("v_fma_f32 v64, v64, v96, v101\n v_fma_f32 v68, v68, v96, v101\n v_fma_f32 v72, v72, v96, v101\n v_fma_f32 v76, v76, v96, v101\n v_fma_f32 v80, v80, v96, v101\n v_fma_f32 v84, v84, v96, v101\n v_fma_f32 v88, v88, v96, v101\n v_fma_f32 v92, v92, v96, v101\n"
                                             "v_fma_f32 v64, v64, v96, v101\n v_fma_f32 v68, v68, v96, v101\n v_fma_f32 v72, v72, v96, v101\n v_fma_f32 v76, v76, v96, v101\n v_fma_f32 v80, v80, v96, v101\n v_fma_f32 v84, v84, v96, v101\n v_fma_f32 v88, v88, v96, v101\n v_fma_f32 v92, v92, v96, v101\n");
#define F(n) "v_add_f32 v" #n ", v" #n ", v97\n"
#define S(n) "v_alignbit_b32 v" #n ", v" #n ", v99, 31\n"
    else if (KIND == K_PAT_FS) BODY1(F(64) S(80) F(65) S(81) F(66) S(82) F(67) S(83) F(68) S(84) F(69) S(85) F(70) S(86) F(71) S(87));
    else if (KIND == K_PAT_FFSS) BODY1(F(64) F(65) S(80) S(81) F(66) F(67) S(82) S(83) F(68) F(69) S(84) S(85) F(70) F(71) S(86) S(87));
    else if (KIND == K_PAT_FFFS) BODY1(F(64) F(65) F(66) S(80) F(67) F(68) F(69) S(81) F(70) F(71) F(72) S(82) F(73) F(74) F(75) S(83));
    else if (KIND == K_PAT_F8S8) BODY1(F(64) F(65) F(66) F(67) F(68) F(69) F(70) F(71) S(80) S(81) S(82) S(83) S(84) S(85) S(86) S(87));
    else if (KIND == K_PAT_F32S32) asm volatile(".rept 4\n .rept 2\n" F(64) F(65) F(66) F(67) F(68) F(69) F(70) F(71) F(72) F(73) F(74) F(75) F(76) F(77) F(78) F(79) ".endr\n"
                                                ".rept 2\n" S(80) S(81) S(82) S(83) S(84) S(85) S(86) S(87) S(88) S(89) S(90) S(91) S(92) S(93) S(94) S(95) ".endr\n .endr\n" ::: CLOB);
    else if (KIND == K_PAT_FS_OFFSET) {
      if (odd_wave) BODY1(S(80) F(64) S(81) F(65) S(82) F(66) S(83) F(67) S(84) F(68) S(85) F(69) S(86) F(70) S(87) F(71));
      else BODY1(F(64) S(80) F(65) S(81) F(66) S(82) F(67) S(83) F(68) S(84) F(69) S(85) F(70) S(86) F(71) S(87));
    }
#undef F
#undef S
  }
  float acc;
  asm volatile("v_add_f32 %0, v64, v65\n v_add_f32 %0, %0, v80\n v_add_f32 %0, %0, v72" : "=v"(acc) :: CLOB);
  if (acc == 12345.678f) out[0] = acc + lds_pad[threadIdx.x];
}

template <int KIND>
static void row(int cus, float* d_out, bool pmc) {
  printf("%-52s", names[KIND]);
  for (int w : {2, 8}) {
    if (pmc && w != 8) continue;
    const int n_waves = cus * 4 * w;
    const size_t lds = std::min<size_t>(65536, (160 * 1024) / (4 * w)) & ~size_t(255);
    const int iters = pmc ? 4000 : std::max(1, (int)(48e6 / ((double)w * 256)));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    if (!pmc) hipLaunchKernelGGL(k<KIND>, dim3(n_waves), dim3(64), lds, 0, d_out, iters / 4 + 1, 1.0001f, 0x9e3779b9u);
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<KIND>, dim3(n_waves), dim3(64), lds, 0, d_out, iters, 1.0001f, 0x9e3779b9u);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    // nominal 2.39 GHz: the clock every valu_ceiling cell measured within 2 % at 2 and 8 waves
    printf(" | w=%d %5.2f cyc/instr (%5.1f ms)", w, ms * 1e-3 * 2.39e9 / ((double)iters * 256 * w), ms);
    CHECK(hipEventDestroy(e0)); CHECK(hipEventDestroy(e1));
  }
  printf("\n");
  fflush(stdout);
}

template <int K0, int K1>
struct all_rows {
  static void go(int cus, float* d_out, bool pmc) { row<K0>(cus, d_out, pmc); all_rows<K0 + 1, K1>::go(cus, d_out, pmc); }
};
template <int K1>
struct all_rows<K1, K1> { static void go(int, float*, bool) {} };

int main(int argc, char** argv) {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  float* d_out;
  CHECK(hipMalloc(&d_out, 4));
  const bool pmc = argc > 1 && !strcmp(argv[1], "pmc");
  printf("device %s, %d CUs; SIMD cycles (at 2.39 GHz) per wave64 instruction, wall clock, 2 and 8 waves per SIMD; kernel k<i> = row i\n", prop.gcnArchName, prop.multiProcessorCount);
  all_rows<0, K_COUNT>::go(prop.multiProcessorCount, d_out, pmc);
  return 0;
}
