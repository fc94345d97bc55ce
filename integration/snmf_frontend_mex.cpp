// snmf_frontend_mex.cpp -- MATLAB MEX shim for the device spectrogram front-end of libsnmf_hip.so
// (C ABI: include/snmf.h, snmf_stft_features_f32 / snmf_mel_features_f32).
//
// Replaces, for the callers that build V before a solve, the feature lines of the reference:
//     [TF_mag, ~] = stft_fft(s, p.framelength, p.frameshift, p.fftlength, DC_bin, p.win_STFT, p.preemph);   run_basis_train.m:60
//     TF_mag = TF_mag(:, any(TF_mag,1)); [TF_mag, ~] = frame_splice(TF_mag, p);                              :61-62
//     TF_mag = TF_mag .^ p.pow + p.nonzerofloor;                                                             :63
//     TF_Mel(...) = melmat * TF_mag(...)                                                                     :70-78
// (run_basis_DNMF.m:13-34 and run_basis_DNMF_Mel.m form Y, X, D the same way).
//
//     TF_mag = snmf_frontend_mex('stft', s, p, DC_bin)        s: samples (double or single vector), p: settings struct
//     TF_Mel = snmf_frontend_mex('mel', TF_mag, melmat, K)    melmat: F_order x (fftlength/2+1) (= mel_matrix(...)'), K = 2*Splice+1
//
// Written against the documented MEX C API; MATLAB is not available in the build container, so
// __graft_entry__.build() only SYNTAX-CHECKS this file against integration/mex_stub/mex.h.  Build:
//     mex -R2018a -I<repo>/include integration/snmf_frontend_mex.cpp -L<repo>/se_snmf_nat_amd -lsnmf_hip
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "mex.h"
#include "snmf.h"

static snmf_ctx* g_ctx = nullptr;

static void at_exit() {
    if (g_ctx) {
        snmf_ctx_destroy(g_ctx);
        g_ctx = nullptr;
    }
}
static void need_ctx() {
    if (g_ctx) return;
    if (snmf_ctx_create(&g_ctx, 0) != SNMF_OK) mexErrMsgIdAndTxt("snmf:device", "%s", snmf_last_error());
    mexLock();
    mexAtExit(at_exit);
}
static double field(const mxArray* s, const char* name) {
    const mxArray* f = mxGetField(s, 0, name);
    if (!f || mxIsEmpty(f)) mexErrMsgIdAndTxt("snmf:field", "Reference to non-existent field '%s'.", name);
    return mxGetScalar(f);
}
static std::vector<float> to_float(const mxArray* a, const char* what) {
    if (!mxIsDouble(a) || mxIsComplex(a)) mexErrMsgIdAndTxt("snmf:type", "%s must be real double", what);
    const size_t n = mxGetNumberOfElements(a);
    const double* d = mxGetDoubles(a);
    std::vector<float> out(n);
    for (size_t i = 0; i < n; ++i) out[i] = (float)d[i];
    return out;
}

void mexFunction(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]) {
    if (snmf_abi_version() != SNMF_ABI_VERSION)  // a stale libsnmf_hip.so must not be driven through newer prototypes
        mexErrMsgIdAndTxt("snmf:abi", "libsnmf_hip.so has ABI version %d, this MEX file was built against %d", snmf_abi_version(), SNMF_ABI_VERSION);
    if (nrhs < 1 || !mxIsChar(prhs[0])) mexErrMsgIdAndTxt("snmf:nargin", "usage: snmf_frontend_mex('stft'|'mel', ...)");
    if (nlhs > 1) mexErrMsgIdAndTxt("snmf:nargout", "one output");
    char cmd[16];
    mxGetString(prhs[0], cmd, sizeof cmd);
    need_ctx();
    if (std::string(cmd) == "stft") {
        if (nrhs != 4 || !mxIsStruct(prhs[2])) mexErrMsgIdAndTxt("snmf:nargin", "TF_mag = snmf_frontend_mex('stft', s, p, DC_bin)");
        const mxArray* p = prhs[2];
        const std::vector<float> s = to_float(prhs[1], "s");
        const mxArray* win = mxGetField(p, 0, "win_STFT");
        if (!win || !mxIsDouble(win)) mexErrMsgIdAndTxt("snmf:field", "Reference to non-existent field 'win_STFT'.");
        snmf_stft_params sp;
        std::memset(&sp, 0, sizeof sp);
        sp.framelength = (int32_t)field(p, "framelength");
        sp.frameshift = (int32_t)field(p, "frameshift");
        sp.fftlength = (int32_t)field(p, "fftlength");
        sp.dcbin = (int32_t)mxGetScalar(prhs[3]);
        sp.splice = (int32_t)field(p, "Splice");
        sp.preemph = field(p, "preemph");
        sp.pow = field(p, "pow");
        sp.nonzerofloor = field(p, "nonzerofloor");
        if (mxGetNumberOfElements(win) != (size_t)sp.framelength) mexErrMsgIdAndTxt("snmf:dim", "win_STFT must have framelength entries");
        sp.window = mxGetDoubles(win);
        const int64_t nfr = snmf_stft_num_frames(&sp, (int64_t)s.size());
        const size_t F = (size_t)(2 * sp.splice + 1) * (size_t)(sp.fftlength / 2 + 1);
        std::vector<float> V(F * (size_t)(nfr > 0 ? nfr : 1));
        int32_t n_out = 0;
        if (snmf_stft_features_f32(g_ctx, &sp, s.data(), (int64_t)s.size(), 0, V.data(), (int64_t)F, 0, &n_out) != SNMF_OK)
            mexErrMsgIdAndTxt("snmf:stft", "%s", snmf_last_error());
        plhs[0] = mxCreateDoubleMatrix(F, (size_t)n_out, mxREAL);
        double* o = mxGetDoubles(plhs[0]);
        for (size_t i = 0; i < F * (size_t)n_out; ++i) o[i] = (double)V[i];
    } else if (std::string(cmd) == "mel") {
        if (nrhs != 4) mexErrMsgIdAndTxt("snmf:nargin", "TF_Mel = snmf_frontend_mex('mel', TF_mag, melmat, K)");
        const std::vector<float> V = to_float(prhs[1], "TF_mag");
        const size_t rows = mxGetM(prhs[1]), T = mxGetN(prhs[1]);
        const size_t M = mxGetM(prhs[2]), n = mxGetN(prhs[2]);
        const int K = (int)mxGetScalar(prhs[3]);
        if (K < 1 || rows != (size_t)K * n) mexErrMsgIdAndTxt("snmf:dim", "TF_mag must have K * size(melmat,2) rows");
        // the C ABI takes melmat row-major (M x n); MATLAB stores it column-major
        const double* mm = mxGetDoubles(prhs[2]);
        std::vector<float> mel(M * n);
        for (size_t i = 0; i < M; ++i)
            for (size_t j = 0; j < n; ++j) mel[i * n + j] = (float)mm[j * M + i];
        std::vector<float> out((size_t)K * M * T);
        if (snmf_mel_features_f32(g_ctx, mel.data(), (int32_t)M, (int32_t)n, K, V.data(), (int64_t)rows, (int32_t)T, out.data(),
                                  (int64_t)((size_t)K * M), 0) != SNMF_OK)
            mexErrMsgIdAndTxt("snmf:mel", "%s", snmf_last_error());
        plhs[0] = mxCreateDoubleMatrix((size_t)K * M, T, mxREAL);
        double* o = mxGetDoubles(plhs[0]);
        for (size_t i = 0; i < out.size(); ++i) o[i] = (double)out[i];
    } else {
        mexErrMsgIdAndTxt("snmf:cmd", "unknown command '%s'", cmd);
    }
}
