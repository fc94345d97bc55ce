// snmf_dnmf_mex.cpp -- MATLAB MEX shim for the device-resident training callers of libsnmf_hip.so
// (C ABI: include/snmf.h, snmf_run_basis_dnmf_audio_f64 / snmf_run_basis_train_audio_f64 / snmf_stft_num_frames).
//
// Replaces whole reference FUNCTIONS rather than single sparse_nmf calls, so that the spectrograms and the activations of
// solve 1 never come back to MATLAB between the solves:
//     B_hat = run_basis_DNMF(x, d, B, p)          run_basis_DNMF.m:1-55      -> integration/run_basis_DNMF.m
//     B_hat = run_basis_DNMF_Mel(x, d, B, p)      run_basis_DNMF_Mel.m:1-95  -> integration/run_basis_DNMF_Mel.m
//     the feature + two-solve block of run_basis_train.m:58-91               -> integration/snmf_basis_train_block.m
//
//     n           = snmf_dnmf_mex('nframes', n_samples, p)
//     [B_hat, ni] = snmf_dnmf_mex('dnmf',  x, d, B, H0, p, melmat)      melmat: [] (DFT) or F_order x (fftlength/2+1) = mel_matrix(...)'
//     [B_hat, ni] = snmf_dnmf_mex('dnmf_multi', Y, X, D, B, H0, p, devices)   formed features, frames sharded over the devices (snmf_run_basis_dnmf_multi_f64)
//     [B_DFT, B_Mel, A_DFT, A_Mel, ni] = snmf_dnmf_mex('train', s_full, sample_idx, H0, p, melmat, DC_bin)
//   H0: (R_x+R_d) x n_frames (resp. r x n_frames) double -- rand(r, n) drawn by the wrapper with MATLAB's generator exactly as
//       src/sparse_nmf.m:112-114,:133-134 would -- or [] to let the engine draw it on the device (snmf_plan_set_h_random).
//   p : the settings struct (framelength, frameshift, fftlength, DCbin, win_STFT, preemph, pow, nonzerofloor, Splice, R_x, R_d,
//       cf / beta, sparsity, max_iter, conv_eps, cost_check, random_seed, domain_DD, alpha_eta, train_Exemplar)
//
// Written against the documented MEX C API; MATLAB is not available in the build container, so __graft_entry__.build() only
// SYNTAX-CHECKS this file against integration/mex_stub/mex.h.  Build:
//     mex -R2018a -I<repo>/include integration/snmf_dnmf_mex.cpp -L<repo>/se_snmf_nat_amd -lsnmf_hip
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "mex.h"
#include "snmf.h"

static snmf_ctx* g_ctx = nullptr;

static void at_exit() {
    snmf_multi_release_cache();  // the device lists' contexts, pinned buffers and gather buffers (opts.devices)
    if (g_ctx) {
        snmf_ctx_destroy(g_ctx);
        g_ctx = nullptr;
    }
}
static void need_ctx() {
    if (g_ctx) return;
    if (snmf_ctx_create(&g_ctx, 0) != SNMF_OK) mexErrMsgIdAndTxt("snmf:device", "%s", snmf_last_error());
    mexLock();  // keeps the context alive between calls: its pinned transfer buffers and second stream are created once
    mexAtExit(at_exit);
}
static double field(const mxArray* s, const char* name) {
    const mxArray* f = mxGetField(s, 0, name);
    if (!f || mxIsEmpty(f)) mexErrMsgIdAndTxt("snmf:field", "Reference to non-existent field '%s'.", name);
    return mxGetScalar(f);
}
static double field_or(const mxArray* s, const char* name, double dflt) {
    const mxArray* f = mxGetField(s, 0, name);
    return (!f || mxIsEmpty(f)) ? dflt : mxGetScalar(f);
}
static std::vector<float> to_float(const mxArray* a, const char* what) {
    if (!mxIsDouble(a) || mxIsComplex(a)) mexErrMsgIdAndTxt("snmf:type", "%s must be real double", what);
    const size_t n = mxGetNumberOfElements(a);
    const double* d = mxGetDoubles(a);
    std::vector<float> out(n);
    for (size_t i = 0; i < n; ++i) out[i] = (float)d[i];
    return out;
}
// melmat (F_order x n, column-major in MATLAB) -> ROW-major float for the C ABI
static std::vector<float> mel_rows(const mxArray* m, int* M, int n_expect) {
    std::vector<float> out;
    *M = 0;
    if (!m || mxIsEmpty(m)) return out;
    if (!mxIsDouble(m) || mxIsComplex(m) || (int)mxGetN(m) != n_expect) mexErrMsgIdAndTxt("snmf:dim", "melmat must be F_order x (fftlength/2+1) double");
    *M = (int)mxGetM(m);
    const double* d = mxGetDoubles(m);
    out.resize((size_t)*M * n_expect);
    for (int j = 0; j < *M; ++j)
        for (int f = 0; f < n_expect; ++f) out[(size_t)j * n_expect + f] = (float)d[(size_t)f * *M + j];
    return out;
}
static void fill_stft(const mxArray* p, std::vector<double>& win, snmf_stft_params* sp, double dcbin) {
    std::memset(sp, 0, sizeof *sp);
    sp->framelength = (int32_t)field(p, "framelength");
    sp->frameshift = (int32_t)field(p, "frameshift");
    sp->fftlength = (int32_t)field(p, "fftlength");
    sp->dcbin = (int32_t)dcbin;
    sp->splice = (int32_t)field_or(p, "Splice", 0);
    sp->preemph = field_or(p, "preemph", 0.0);
    sp->pow = field(p, "pow");
    sp->nonzerofloor = field(p, "nonzerofloor");
    const mxArray* w = mxGetField(p, 0, "win_STFT");
    if (!w || !mxIsDouble(w) || (int32_t)mxGetNumberOfElements(w) != sp->framelength)
        mexErrMsgIdAndTxt("snmf:field", "p.win_STFT must hold p.framelength doubles");
    win.assign(mxGetDoubles(w), mxGetDoubles(w) + sp->framelength);
    sp->window = win.data();
}
// the solver fields of the settings struct as src/sparse_nmf.m:79-110 reads them; p.cost_check has no default (:260)
static void fill_solver(const mxArray* p, snmf_params* q) {
    std::memset(q, 0, sizeof *q);
    char cf[8] = "kl";
    if (const mxArray* c = mxGetField(p, 0, "cf")) mxGetString(c, cf, sizeof cf);
    q->beta = !std::strcmp(cf, "is") ? 0.0 : !std::strcmp(cf, "kl") ? 1.0 : !std::strcmp(cf, "ed") ? 2.0 : field_or(p, "beta", 1.0);
    q->max_iter = (int32_t)field_or(p, "max_iter", 100);
    q->conv_eps = field_or(p, "conv_eps", 0.0);
    q->cost_check = field(p, "cost_check") != 0.0;
    q->floor_v = 1;
    const mxArray* s = mxGetField(p, 0, "sparsity");
    if (s && mxGetNumberOfElements(s) > 1) mexErrMsgIdAndTxt("snmf:dim", "this caller needs a scalar p.sparsity");
    q->sparsity_kind = SNMF_SPARSITY_SCALAR;
    q->sparsity_scalar = field_or(p, "sparsity", 0.0);
}

// p.random_seed as the key of the device generator.  src/sparse_nmf.m:112 re-seeds only for random_seed > 0 ("<= 0: keep the
// generator's state"), which a stateless counter-based generator cannot mean: with H0 = [] such a seed is an error here, and the
// double is converted through int64 (a negative double -> uint64 is undefined behaviour).
static uint64_t device_seed(const mxArray* p, bool h0_given) {
    const double sd = field_or(p, "random_seed", 1);
    if (h0_given) return 1;  // (not used: the caller's own draws are the initial activations)
    if (!(sd >= 1.0)) mexErrMsgIdAndTxt("snmf:field", "H0 = [] needs p.random_seed >= 1: the device generator is keyed by it (src/sparse_nmf.m:112: <= 0 means 'do not re-seed')");
    return (uint64_t)(int64_t)sd;
}

void mexFunction(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]) {
    if (snmf_abi_version() != SNMF_ABI_VERSION)  // a stale libsnmf_hip.so must not be driven through newer prototypes
        mexErrMsgIdAndTxt("snmf:abi", "libsnmf_hip.so has ABI version %d, this MEX file was built against %d", snmf_abi_version(), SNMF_ABI_VERSION);
    if (nrhs < 1 || !mxIsChar(prhs[0])) mexErrMsgIdAndTxt("snmf:nargin", "usage: snmf_dnmf_mex('nframes'|'dnmf'|'train', ...)");
    char cmd[16];
    mxGetString(prhs[0], cmd, sizeof cmd);
    std::vector<double> win;
    snmf_stft_params sp;
    if (!std::strcmp(cmd, "nframes")) {
        if (nrhs != 3 || !mxIsStruct(prhs[2])) mexErrMsgIdAndTxt("snmf:nargin", "usage: n = snmf_dnmf_mex('nframes', n_samples, p)");
        fill_stft(prhs[2], win, &sp, field_or(prhs[2], "DCbin", 1));
        plhs[0] = mxCreateDoubleScalar((double)snmf_stft_num_frames(&sp, (int64_t)mxGetScalar(prhs[1])));
        return;
    }
    if (!std::strcmp(cmd, "dnmf_multi")) {  // (no context of this file's own: the library keeps one team of contexts per device list)
        if (nrhs != 8 || !mxIsStruct(prhs[6])) mexErrMsgIdAndTxt("snmf:nargin", "usage: [B_hat, n_iter] = snmf_dnmf_mex('dnmf_multi', Y, X, D, B, H0, p, devices)");
        const mxArray *Y = prhs[1], *X = prhs[2], *D = prhs[3], *B = prhs[4], *H0 = prhs[5], *p = prhs[6], *dv = prhs[7];
        const int R_x = (int)field(p, "R_x"), R_d = (int)field(p, "R_d");
        snmf_params q;
        fill_solver(p, &q);
        q.F = (int32_t)mxGetM(Y);
        q.T = (int32_t)mxGetN(Y);
        q.r = R_x + R_d;
        const mxArray* mats[3] = {Y, X, D};
        for (const mxArray* M_ : mats)
            if (!mxIsDouble(M_) || mxIsComplex(M_) || (int)mxGetM(M_) != q.F || (int)mxGetN(M_) != q.T)
                mexErrMsgIdAndTxt("snmf:dim", "Y, X and D must be real double matrices of one size");
        if (!mxIsDouble(B) || mxIsComplex(B) || (int)mxGetM(B) != q.F || (int)mxGetN(B) != q.r)
            mexErrMsgIdAndTxt("snmf:dim", "B must be %d x (R_x + R_d = %d) double", q.F, q.r);
        const double* h0 = nullptr;
        if (!mxIsEmpty(H0)) {
            if (!mxIsDouble(H0) || (int)mxGetM(H0) != q.r || (int)mxGetN(H0) != q.T) mexErrMsgIdAndTxt("snmf:dim", "H0 must be %d x %d double or []", q.r, q.T);
            h0 = mxGetDoubles(H0);
        }
        const size_t nd = mxGetNumberOfElements(dv);
        if (!mxIsDouble(dv) || nd < 1 || nd > 16) mexErrMsgIdAndTxt("snmf:dim", "devices must be a double vector of 1..16 zero-based device ordinals");
        std::vector<int32_t> devs(nd);
        for (size_t i = 0; i < nd; ++i) devs[i] = (int32_t)mxGetDoubles(dv)[i];
        const double seed_d = field_or(p, "random_seed", 1);
        if (!h0 && !(seed_d >= 1.0)) mexErrMsgIdAndTxt("snmf:field", "the device generator needs p.random_seed >= 1 (src/sparse_nmf.m:112: <= 0 means 'do not re-seed', which a stateless generator cannot mean)");
        plhs[0] = mxCreateDoubleMatrix((mwSize)q.F, (mwSize)q.r, mxREAL);
        int32_t nit[3] = {0, 0, 0};
        const int st = snmf_run_basis_dnmf_multi_f64(devs.data(), (int32_t)nd, &q, R_x, R_d, mxGetDoubles(Y), q.F, mxGetDoubles(X), q.F,
                                                     mxGetDoubles(D), q.F, mxGetDoubles(B), q.F, h0, (uint64_t)(int64_t)(seed_d >= 1.0 ? seed_d : 1.0),
                                                     mxGetDoubles(plhs[0]), q.F, nullptr, q.r, nit);
        if (st != SNMF_OK) mexErrMsgIdAndTxt("snmf:solve", "%s", snmf_last_error());
        if (nlhs > 1) {
            plhs[1] = mxCreateDoubleMatrix(1, 3, mxREAL);
            for (int i = 0; i < 3; ++i) mxGetDoubles(plhs[1])[i] = nit[i];
        }
        return;
    }
    need_ctx();
    if (!std::strcmp(cmd, "dnmf")) {
        if (nrhs != 7 || !mxIsStruct(prhs[5])) mexErrMsgIdAndTxt("snmf:nargin", "usage: [B_hat, n_iter] = snmf_dnmf_mex('dnmf', x, d, B, H0, p, melmat)");
        const mxArray *B = prhs[3], *H0 = prhs[4], *p = prhs[5];
        const std::vector<float> x = to_float(prhs[1], "x"), d = to_float(prhs[2], "d");
        fill_stft(p, win, &sp, field(p, "DCbin"));
        int M = 0;
        const int nb = sp.fftlength / 2 + 1, K = 2 * sp.splice + 1;
        const std::vector<float> mel = mel_rows(prhs[6], &M, nb);
        const int R_x = (int)field(p, "R_x"), R_d = (int)field(p, "R_d");
        snmf_params q;
        fill_solver(p, &q);
        q.F = M ? K * M : K * nb;
        q.T = (int32_t)snmf_stft_num_frames(&sp, (int64_t)(x.size() < d.size() ? x.size() : d.size()));
        q.r = R_x + R_d;
        if (!mxIsDouble(B) || mxIsComplex(B) || (int)mxGetM(B) != q.F || (int)mxGetN(B) != q.r)
            mexErrMsgIdAndTxt("snmf:dim", "B must be %d x (R_x + R_d = %d) double", q.F, q.r);
        const double* h0 = nullptr;
        if (!mxIsEmpty(H0)) {
            if (!mxIsDouble(H0) || (int)mxGetM(H0) != q.r || (int)mxGetN(H0) != q.T) mexErrMsgIdAndTxt("snmf:dim", "H0 must be %d x %d double or []", q.r, q.T);
            h0 = mxGetDoubles(H0);
        }
        plhs[0] = mxCreateDoubleMatrix((mwSize)q.F, (mwSize)q.r, mxREAL);
        int32_t nit[3] = {0, 0, 0};
        const int st = snmf_run_basis_dnmf_audio_f64(g_ctx, &q, &sp, R_x, R_d, x.data(), (int64_t)x.size(), d.data(), (int64_t)d.size(),
                                                     M ? mel.data() : nullptr, M, mxGetDoubles(B), q.F, h0,
                                                     device_seed(p, h0 != nullptr), mxGetDoubles(plhs[0]), q.F, nullptr, q.r, nit);
        if (st != SNMF_OK) mexErrMsgIdAndTxt("snmf:solve", "%s", snmf_last_error());
        if (nlhs > 1) {
            plhs[1] = mxCreateDoubleMatrix(1, 3, mxREAL);
            for (int i = 0; i < 3; ++i) mxGetDoubles(plhs[1])[i] = nit[i];
        }
        return;
    }
    if (!std::strcmp(cmd, "train")) {
        if (nrhs != 7 || !mxIsStruct(prhs[4])) mexErrMsgIdAndTxt("snmf:nargin", "usage: [B_DFT,B_Mel,A_DFT,A_Mel,n_iter] = snmf_dnmf_mex('train', s_full, sample_idx, H0, p, melmat, DC_bin)");
        const mxArray *idx = prhs[2], *H0 = prhs[3], *p = prhs[4];
        const std::vector<float> s = to_float(prhs[1], "s_full");
        fill_stft(p, win, &sp, mxGetScalar(prhs[6]));
        int M = 0;
        const int nb = sp.fftlength / 2 + 1, K = 2 * sp.splice + 1;
        const std::vector<float> mel = mel_rows(prhs[5], &M, nb);
        if (!M) mexErrMsgIdAndTxt("snmf:dim", "melmat is required (run_basis_train.m:70-78 always forms TF_Mel)");
        snmf_params q;
        const bool exemplar = field_or(p, "train_Exemplar", 0) != 0.0;
        if (exemplar) {  // :84: no solve at all, the solver fields are not read
            std::memset(&q, 0, sizeof q);
            q.beta = 1.0;
            q.max_iter = 1;
            q.floor_v = 1;
        } else {
            fill_solver(p, &q);
        }
        q.F = K * nb;
        q.T = (int32_t)snmf_stft_num_frames(&sp, (int64_t)s.size());
        q.r = (int32_t)mxGetNumberOfElements(idx);
        if (!mxIsDouble(idx) || q.r < 1) mexErrMsgIdAndTxt("snmf:dim", "sample_idx must be a double vector of 1-based frame indices");
        std::vector<int64_t> i0((size_t)q.r);
        for (int j = 0; j < q.r; ++j) i0[j] = (int64_t)mxGetDoubles(idx)[j] - 1;  // randsample is 1-based (:81)
        const double* h0 = nullptr;
        if (!mxIsEmpty(H0)) {
            if (!mxIsDouble(H0) || (int)mxGetM(H0) != q.r || (int)mxGetN(H0) != q.T) mexErrMsgIdAndTxt("snmf:dim", "H0 must be %d x %d double or []", q.r, q.T);
            h0 = mxGetDoubles(H0);
        }
        plhs[0] = mxCreateDoubleMatrix((mwSize)q.F, (mwSize)q.r, mxREAL);
        mxArray* BM = mxCreateDoubleMatrix((mwSize)(K * M), (mwSize)q.r, mxREAL);
        mxArray* AD = exemplar ? mxCreateDoubleScalar(0.0) : mxCreateDoubleMatrix((mwSize)q.r, (mwSize)q.T, mxREAL);  // :95-96
        mxArray* AM = exemplar ? mxCreateDoubleScalar(0.0) : mxCreateDoubleMatrix((mwSize)q.r, (mwSize)q.T, mxREAL);
        int32_t nit[2] = {0, 0};
        const double dd = field_or(p, "domain_DD", 0) != 0.0 ? field(p, "alpha_eta") : -1.0;  // :64-67
        const int st = snmf_run_basis_train_audio_f64(g_ctx, &q, &sp, dd, mel.data(), M, s.data(), (int64_t)s.size(), i0.data(), exemplar ? 1 : 0,
                                                      h0, device_seed(p, h0 != nullptr || exemplar), mxGetDoubles(plhs[0]),
                                                      exemplar ? nullptr : mxGetDoubles(AD), mxGetDoubles(BM), exemplar ? nullptr : mxGetDoubles(AM), nit);
        if (st != SNMF_OK) {
            mxDestroyArray(BM);
            mxDestroyArray(AD);
            mxDestroyArray(AM);
            mexErrMsgIdAndTxt("snmf:solve", "%s", snmf_last_error());
        }
        if (nlhs > 1) plhs[1] = BM; else mxDestroyArray(BM);
        if (nlhs > 2) plhs[2] = AD; else mxDestroyArray(AD);
        if (nlhs > 3) plhs[3] = AM; else mxDestroyArray(AM);
        if (nlhs > 4) {
            plhs[4] = mxCreateDoubleMatrix(1, 2, mxREAL);
            mxGetDoubles(plhs[4])[0] = nit[0];
            mxGetDoubles(plhs[4])[1] = nit[1];
        }
        return;
    }
    mexErrMsgIdAndTxt("snmf:cmd", "unknown command '%s'", cmd);
}
