// Test-only C entry points over phyloformer_amd/csrc/pf_host_prep.h (the host-side operand preparation of
// libphyloformer_amd.so), so that tests/native/fuzz_host.py can drive it under AddressSanitizer / UBSan
// (compiled with g++ -fsanitize=address,undefined; no HIP anywhere in this translation unit).
#include "../../phyloformer_amd/csrc/pf_host_prep.h"

using namespace pfhost;

extern "C" {

// out: (Mpad / 32) * (K / 16) * 2 * 64 * 8 uint16
void t_pack_frags(const float* W, int M, int K, int Mpad, uint16_t* out) { pack_frags(W, M, K, Mpad, out); }
// the host's float -> IEEE half -> float conversions (what packs the weights; the device's v_cvt_pk_f16_f32 agrees
// bit for bit, tools/f16_probe.hip)
void t_f2h(const float* in, uint16_t* out, int n) { for (int i = 0; i < n; ++i) out[i] = f2h(in[i]); }
void t_h2f(const uint16_t* in, float* out, int n) { for (int i = 0; i < n; ++i) out[i] = h2f(in[i]); }
int t_operand_format() { return PF_F16; }
// tail: (FRAG_END - FRAG_WV) * 8 uint16, wv_lo: WVLO_FRAGS * 8 uint16
void t_pack_row_stats(const float* wv, const float* wq, const float* wk, uint16_t* tail, uint16_t* wv_lo) {
    pack_row_stats(wv, wq, wk, tail, wv_lo);
}
int t_frag_sizes(int* tail_u16, int* wvlo_u16, int* frag_qk_off_u16) {
    *tail_u16 = (FRAG_END - FRAG_WV) * 8;
    *wvlo_u16 = WVLO_FRAGS * 8;
    *frag_qk_off_u16 = (FRAG_QK - FRAG_WV) * 8;
    return 0;
}
// Wf: M * K floats, bf: M floats
void t_fold(const float* W, const float* bias, const float* g, const float* beta, int M, int K, float* Wf, float* bf) {
    std::vector<float> a, b;
    fold(W, bias, g, beta, M, K, a, b);
    std::memcpy(Wf, a.data(), a.size() * sizeof(float));
    std::memcpy(bf, b.data(), b.size() * sizeof(float));
}
// table: [22][64]; attention weights of block 0's row attention; out: [484][72]
void t_build_pair_table(const float* table, const float* g, const float* b, const float* wq, const float* bq,
                        const float* wk, const float* bk, const float* wv, float* out) {
    AttnHost r{};
    r.g = g; r.b = b; r.wq = wq; r.bq = bq; r.wk = wk; r.bk = bk; r.wv = wv;
    std::vector<float> t;
    build_pair_table(table, r, t);
    std::memcpy(out, t.data(), t.size() * sizeof(float));
}
void t_tile_plan(int P, int Lloc, int tile_force, int* flat, int* nt_aln, int* slots_aln) {
    const TilePlan t = tile_plan_core(P, Lloc, tile_force);
    *flat = t.flat; *nt_aln = t.nt_aln; *slots_aln = t.slots_aln;
}
void t_colstats_plan(int B, int P, int Lloc, int fine_opt, int* G, int* sub, int* S, int* fine) {
    const ColPlan c = colstats_plan_core(B, P, Lloc, fine_opt);
    *G = c.G; *sub = c.sub; *S = c.S; *fine = c.fine;
}

}  // extern "C"
