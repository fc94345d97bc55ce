// snmf_online_mex.cpp -- MATLAB MEX shim over the online-separation entry points of libsnmf_hip.so
// (include/snmf.h: snmf_online_*).
//
// The binding a maintainer of lordet01/SE_SNMF_NAT adds so that the frame loop of
// src/NTF_sep_event_RT.m:54-135 -- init_buff + one bnmf_sep_event_RT_IS16 call per hop -- runs on an
// MI355X (integration/NTF_sep_event_RT.m is the MATLAB side).  Written against the documented MEX C API;
// MATLAB is not available in the build container or on the GPU box, so this file is NOT compiled by
// __graft_entry__.build(); the same C ABI is exercised from Python (se_snmf_nat_amd/online.py).
//
// Build:  mex -R2018a -I<repo>/include integration/snmf_online_mex.cpp -L<repo>/se_snmf_nat_amd -lsnmf_hip
//
// Calls:
//   h = snmf_online_mex('create', B_DFT_x, B_DFT_d, H0, Ad_blk0, p)   p = the settings struct (global p)
//   x_tilde_int16 = snmf_online_mex('process', h, pcm, flush)
//   B_DFT_d = snmf_online_mex('basis', h, F, R_d)          g.B_DFT_d, saved to B_D_u.mat by src/NTF_sep_event_RT.m:138-140
//   snmf_online_mex('destroy', h)
// H0 = rand(R_x+R_d,1) after rand('seed',p.random_seed) and Ad_blk0 = rand(p.R_a,p.m_a) are drawn by the
// MATLAB wrapper with MATLAB's own generator (src/sparse_nmf.m:112-114,:133-134; src/init_buff.m:39).
#include <cmath>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "mex.h"
#include "snmf.h"

static snmf_ctx* g_ctx = nullptr;
static std::vector<snmf_online*> g_handles;

static void at_exit() {
    for (snmf_online* o : g_handles)
        if (o) snmf_online_destroy(o);
    g_handles.clear();
    if (g_ctx) {
        snmf_ctx_destroy(g_ctx);
        g_ctx = nullptr;
    }
}

static double fld(const mxArray* p, const char* name, double dflt, bool required = false) {
    const mxArray* f = mxGetField(p, 0, name);
    if (!f || mxIsEmpty(f)) {
        if (required) mexErrMsgIdAndTxt("snmf:field", "Reference to non-existent field '%s'.", name);
        return dflt;
    }
    return mxGetScalar(f);
}

static std::vector<float> to_f32(const mxArray* a, const char* what) {
    if (!mxIsDouble(a) || mxIsComplex(a)) mexErrMsgIdAndTxt("snmf:type", "%s must be real double", what);
    const size_t n = mxGetNumberOfElements(a);
    const double* d = mxGetDoubles(a);
    std::vector<float> out(n);
    for (size_t i = 0; i < n; ++i) out[i] = (float)d[i];
    return out;
}

static snmf_online* handle_of(const mxArray* a) {
    const size_t i = (size_t)mxGetScalar(a);
    if (i < 1 || i > g_handles.size() || !g_handles[i - 1]) mexErrMsgIdAndTxt("snmf:handle", "invalid separator handle");
    return g_handles[i - 1];
}

void mexFunction(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]) {
    if (snmf_abi_version() != SNMF_ABI_VERSION)  // a stale libsnmf_hip.so must not be driven through newer prototypes
        mexErrMsgIdAndTxt("snmf:abi", "libsnmf_hip.so has ABI version %d, this MEX file was built against %d", snmf_abi_version(), SNMF_ABI_VERSION);
    if (nrhs < 1 || !mxIsChar(prhs[0])) mexErrMsgIdAndTxt("snmf:usage", "first argument: 'create' | 'process' | 'destroy'");
    char cmd[16];
    mxGetString(prhs[0], cmd, sizeof cmd);
    if (!g_ctx) {
        if (snmf_ctx_create(&g_ctx, 0) != SNMF_OK) mexErrMsgIdAndTxt("snmf:device", "%s", snmf_last_error());
        mexAtExit(at_exit);
        mexLock();
    }
    if (!strcmp(cmd, "create")) {
        if (nrhs != 6) mexErrMsgIdAndTxt("snmf:usage", "create: B_DFT_x, B_DFT_d, H0, Ad_blk0, p");
        const mxArray* p = prhs[5];
        if (!mxIsStruct(p)) mexErrMsgIdAndTxt("snmf:type", "p must be a struct");
        char mode[8] = "DFT", meth[8] = "MMSE", cf[8] = "kl";
        if (const mxArray* f = mxGetField(p, 0, "B_sep_mode")) mxGetString(f, mode, sizeof mode);
        if (const mxArray* f = mxGetField(p, 0, "ENHANCE_METHOD")) mxGetString(f, meth, sizeof meth);
        if (const mxArray* f = mxGetField(p, 0, "cf")) mxGetString(f, cf, sizeof cf);
        if ((strcmp(mode, "DFT") && strcmp(mode, "Mel")) || fld(p, "Splice", 0) != 0 || fld(p, "blk_len_sep", 1) != 1)
            mexErrMsgIdAndTxt("snmf:unsupported", "only Splice=0, blk_len_sep=1, B_sep_mode 'DFT' or 'Mel' (then call 'set_mel')");
        snmf_online_params q;
        std::memset(&q, 0, sizeof q);
        q.fftlength = (int32_t)fld(p, "fftlength", 0, true);
        q.framelength = (int32_t)fld(p, "framelength", 0, true);
        q.frameshift = (int32_t)fld(p, "frameshift", 0, true);
        q.dcbin = (int32_t)fld(p, "DCbin", 0, true);
        q.dcbin_back = (int32_t)fld(p, "DCbin_back", q.dcbin);
        q.delay = (int32_t)fld(p, "delay", 0, true);
        q.preemph = fld(p, "preemph", 0.0);
        q.pow = fld(p, "pow", 2.0);
        q.nonzerofloor = fld(p, "nonzerofloor", 1e-9);
        q.overlapscale = fld(p, "overlapscale", 0, true);
        q.R_x = (int32_t)mxGetN(prhs[1]);
        q.R_d = (int32_t)mxGetN(prhs[2]);
        q.beta_div = !strcmp(cf, "is") ? 0.0 : !strcmp(cf, "kl") ? 1.0 : !strcmp(cf, "ed") ? 2.0 : fld(p, "beta_div", 1.0);
        q.sparsity = fld(p, "sparsity", 0.0);
        q.max_iter = (int32_t)fld(p, "max_iter", 100);
        q.cost_check = fld(p, "cost_check", 0, true) != 0;  // src/sparse_nmf.m:260
        q.conv_eps = fld(p, "conv_eps", 0.0);
        q.enhance_method = !strcmp(meth, "Wiener") ? 0 : 1;
        q.init_N_len = (int32_t)fld(p, "init_N_len", 0);
        q.alpha_eta = fld(p, "alpha_eta", 0.4);
        q.alpha_d = fld(p, "alpha_d", 0.6);
        q.beta = fld(p, "beta", 1.0);
        q.beta_max = fld(p, "beta_max", 1000.0);
        q.blk_sparse = fld(p, "blk_sparse", 0) != 0;
        q.P_len_k = (int32_t)fld(p, "P_len_k", 60);
        q.P_len_l = (int32_t)fld(p, "P_len_l", 20);
        q.blk_gap = (int32_t)fld(p, "blk_gap", 3);
        q.alpha_p = fld(p, "alpha_p", 0.4);
        q.adapt_train_N = fld(p, "adapt_train_N", 0) != 0;
        q.R_a = (int32_t)fld(p, "R_a", 1);
        q.m_a = (int32_t)fld(p, "m_a", 1);
        q.overlap_m_a = fld(p, "overlap_m_a", 0.01);
        q.Ar_up = fld(p, "Ar_up", 1.0);
        q.class_outputs = 0;
        q.basis_update_N = fld(p, "basis_update_N", 0) != 0;
        q.basis_update_E = fld(p, "basis_update_E", 0) != 0;
        const std::vector<float> Bx = to_f32(prhs[1], "B_DFT_x"), Bd = to_f32(prhs[2], "B_DFT_d"), H0 = to_f32(prhs[3], "H0"),
                                 Ad = to_f32(prhs[4], "Ad_blk0");
        const mxArray *ws = mxGetField(p, 0, "win_STFT"), *wi = mxGetField(p, 0, "win_ISTFT");
        if (!ws || !wi) mexErrMsgIdAndTxt("snmf:field", "p.win_STFT / p.win_ISTFT missing");
        const std::vector<float> w1 = to_f32(ws, "win_STFT"), w2 = to_f32(wi, "win_ISTFT");
        snmf_online* o = nullptr;
        if (snmf_online_create(g_ctx, &q, Bx.data(), Bd.data(), H0.data(), Ad.data(), w1.data(), w2.data(), &o) != SNMF_OK)
            mexErrMsgIdAndTxt("snmf:create", "%s", snmf_last_error());
        g_handles.push_back(o);
        plhs[0] = mxCreateDoubleScalar((double)g_handles.size());
    } else if (!strcmp(cmd, "set_mel")) {
        // snmf_online_mex('set_mel', h, melmat, B_Mel_x, B_Mel_d, MelConv): melmat = g.melmat (F_order x F), init_buff.m:46
        if (nrhs != 6) mexErrMsgIdAndTxt("snmf:usage", "set_mel: handle, melmat, B_Mel_x, B_Mel_d, MelConv");
        snmf_online* o = handle_of(prhs[1]);
        const mwSize n1 = mxGetM(prhs[2]), F = mxGetN(prhs[2]);
        const double* mm = mxGetDoubles(prhs[2]);
        std::vector<float> mr((size_t)n1 * F);  // MATLAB is column-major, the C ABI wants the rows contiguous
        for (mwSize m = 0; m < n1; ++m)
            for (mwSize f = 0; f < F; ++f) mr[(size_t)m * F + f] = (float)mm[(size_t)f * n1 + m];
        const std::vector<float> bx = to_f32(prhs[3], "B_Mel_x"), bd = to_f32(prhs[4], "B_Mel_d");
        if (snmf_online_set_mel(o, (int32_t)n1, mxGetScalar(prhs[5]) != 0, mr.data(), bx.data(), bd.data()) != SNMF_OK)
            mexErrMsgIdAndTxt("snmf:set_mel", "%s", snmf_last_error());
    } else if (!strcmp(cmd, "process")) {
        if (nrhs != 4) mexErrMsgIdAndTxt("snmf:usage", "process: handle, pcm, flush");
        snmf_online* o = handle_of(prhs[1]);
        const std::vector<float> pcm = to_f32(prhs[2], "pcm");
        const int flush = mxGetScalar(prhs[3]) != 0;
        const int64_t cap = (int64_t)pcm.size() + 64 * 4096;
        std::vector<int16_t> out((size_t)cap);
        int64_t n = 0;
        if (snmf_online_process_f32(o, pcm.data(), (int64_t)pcm.size(), flush, nullptr, out.data(), nullptr, nullptr, cap, &n) != SNMF_OK)
            mexErrMsgIdAndTxt("snmf:process", "%s", snmf_last_error());
        plhs[0] = mxCreateNumericMatrix((mwSize)n, 1, mxINT16_CLASS, mxREAL);
        std::memcpy(mxGetInt16s(plhs[0]), out.data(), (size_t)n * 2);
    } else if (!strcmp(cmd, "basis")) {
        if (nrhs != 4) mexErrMsgIdAndTxt("snmf:usage", "basis: handle, F, R_d");
        snmf_online* o = handle_of(prhs[1]);
        const mwSize F = (mwSize)mxGetScalar(prhs[2]), Rd = (mwSize)mxGetScalar(prhs[3]);
        std::vector<float> B((size_t)F * Rd);
        if (snmf_online_get_basis_f32(o, B.data(), (int64_t)F) != SNMF_OK) mexErrMsgIdAndTxt("snmf:basis", "%s", snmf_last_error());
        plhs[0] = mxCreateDoubleMatrix(F, Rd, mxREAL);
        double* d = mxGetDoubles(plhs[0]);
        for (size_t i = 0; i < B.size(); ++i) d[i] = (double)B[i];
    } else if (!strcmp(cmd, "destroy")) {
        if (nrhs != 2) mexErrMsgIdAndTxt("snmf:usage", "destroy: handle");
        const size_t i = (size_t)mxGetScalar(prhs[1]);
        if (i >= 1 && i <= g_handles.size() && g_handles[i - 1]) {
            snmf_online_destroy(g_handles[i - 1]);
            g_handles[i - 1] = nullptr;
        }
    } else {
        mexErrMsgIdAndTxt("snmf:usage", "unknown command '%s'", cmd);
    }
}
