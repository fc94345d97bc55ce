// sparse_nmf_mex.cpp -- MATLAB MEX shim over libsnmf_hip.so (C ABI: include/snmf.h).
//
// This is the binding a maintainer of lordet01/SE_SNMF_NAT adds to make src/sparse_nmf.m /
// src/sparse_nmf_GPU.m run on an MI355X.  It is written against the documented MEX C API
// (mex.h / matrix.h).  MATLAB is not available in the build container or on the GPU box, so
// __graft_entry__.build() only SYNTAX-CHECKS this file against a stub of those prototypes
// (integration/mex_stub/mex.h); the same C ABI is exercised by the Python ctypes binding
// (se_snmf_nat_amd/_lib.py), which is what the tests drive.
//
// Build (on a machine with MATLAB + ROCm):
//     mex -R2018a -I<repo>/include integration/sparse_nmf_mex.cpp -L<repo>/se_snmf_nat_amd -lsnmf_hip
//
// MATLAB-side call (made by integration/sparse_nmf.m after it has applied the defaults of
// src/sparse_nmf.m:75-164 and drawn the random initial factors with MATLAB's own RNG):
//     [w, h, div, cost, n_iter] = sparse_nmf_mex(v, w0, h0, sparsity, opts)
//   v        F x T double          (src/sparse_nmf.m:71-72)
//   w0       F x r double          init_w                     (:116-131)
//   h0       r x T double          init_h                     (:133-140)
//   sparsity scalar | r x 1 | r x T double                    (:150-155)
//   opts     struct: beta, max_iter, conv_eps, cost_check, floor_v (0 for the _GPU variant),
//            w_update_ind (r x 1 logical), h_update_ind (r x 1 logical), device (0-based),
//            devices (vector of 0-based device ordinals, optional): the frames are sharded over these GPUs inside
//            this one MATLAB process (snmf_sparse_nmf_multi_f64: one exchange of the W statistics per iteration)
#include <cstdint>
#include <cstring>
#include <vector>

#include "mex.h"
#include "snmf.h"

static snmf_ctx* g_ctx = nullptr;
static int g_device = -1;

static void at_exit() {
    snmf_multi_release_cache();  // the device lists' contexts, pinned buffers and gather buffers (opts.devices)
    if (g_ctx) {
        snmf_ctx_destroy(g_ctx);
        g_ctx = nullptr;
    }
}

static double opt_scalar(const mxArray* s, const char* name, double dflt) {
    const mxArray* f = mxGetField(s, 0, name);
    if (!f || mxIsEmpty(f)) return dflt;
    return mxGetScalar(f);
}

static void check_real_double_2d(const mxArray* a, const char* what) {
    if (!mxIsDouble(a) || mxIsComplex(a) || mxGetNumberOfDimensions(a) != 2)
        mexErrMsgIdAndTxt("snmf:type", "%s must be a real double matrix", what);
}

static void fill_mask(const mxArray* opts, const char* name, size_t r, std::vector<uint8_t>& out) {
    out.assign(r, 1);  // src/sparse_nmf.m:142-148: default true(r,1)
    const mxArray* f = mxGetField(opts, 0, name);
    if (!f || mxIsEmpty(f)) return;
    if (mxGetNumberOfElements(f) != r) mexErrMsgIdAndTxt("snmf:dim", "%s must have r entries", name);
    if (mxIsLogical(f)) {
        const mxLogical* p = mxGetLogicals(f);
        for (size_t i = 0; i < r; ++i) out[i] = p[i] ? 1 : 0;
    } else {
        const double* p = mxGetDoubles(f);
        for (size_t i = 0; i < r; ++i) out[i] = p[i] != 0.0;
    }
}

void mexFunction(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]) {
    if (snmf_abi_version() != SNMF_ABI_VERSION)  // a stale libsnmf_hip.so must not be driven through newer prototypes
        mexErrMsgIdAndTxt("snmf:abi", "libsnmf_hip.so has ABI version %d, this MEX file was built against %d", snmf_abi_version(), SNMF_ABI_VERSION);
    if (nrhs != 5) mexErrMsgIdAndTxt("snmf:nargin", "usage: [w,h,div,cost,n_iter] = sparse_nmf_mex(v,w0,h0,sparsity,opts)");
    if (nlhs > 5) mexErrMsgIdAndTxt("snmf:nargout", "too many outputs");
    const mxArray *v = prhs[0], *w0 = prhs[1], *h0 = prhs[2], *sp = prhs[3], *opts = prhs[4];
    check_real_double_2d(v, "v");
    check_real_double_2d(w0, "init_w");
    check_real_double_2d(h0, "init_h");
    check_real_double_2d(sp, "sparsity");
    if (!mxIsStruct(opts)) mexErrMsgIdAndTxt("snmf:type", "opts must be a struct");
    const size_t F = mxGetM(v), T = mxGetN(v), r = mxGetN(w0);
    if (mxGetM(w0) != F) mexErrMsgIdAndTxt("snmf:dim", "init_w must have size(v,1) rows");
    if (mxGetM(h0) != r || mxGetN(h0) != T) mexErrMsgIdAndTxt("snmf:dim", "init_h must be r x size(v,2)");

    const int device = (int)opt_scalar(opts, "device", 0);
    if (!g_ctx || g_device != device) {
        at_exit();
        if (snmf_ctx_create(&g_ctx, device) != SNMF_OK) mexErrMsgIdAndTxt("snmf:device", "%s", snmf_last_error());
        g_device = device;
        mexLock();  // keep the context (device buffers, kernels) alive between calls
        mexAtExit(at_exit);
    }

    snmf_params p;
    std::memset(&p, 0, sizeof p);
    p.F = (int32_t)F;
    p.T = (int32_t)T;
    p.r = (int32_t)r;
    p.beta = opt_scalar(opts, "beta", 1.0);
    p.max_iter = (int32_t)opt_scalar(opts, "max_iter", 100);
    p.conv_eps = opt_scalar(opts, "conv_eps", 0.0);
    p.cost_check = opt_scalar(opts, "cost_check", 1.0) != 0.0;
    p.floor_v = opt_scalar(opts, "floor_v", 1.0) != 0.0;
    std::vector<uint8_t> wi, hi;
    fill_mask(opts, "w_update_ind", r, wi);
    fill_mask(opts, "h_update_ind", r, hi);
    p.w_update_ind = wi.data();
    p.h_update_ind = hi.data();
    const double* sparsity = nullptr;
    const size_t ns = mxGetNumberOfElements(sp);
    if (ns == 1) {
        p.sparsity_kind = SNMF_SPARSITY_SCALAR;  // :151-152
        p.sparsity_scalar = mxGetScalar(sp);
    } else if (mxGetN(sp) == 1 && mxGetM(sp) == r) {
        p.sparsity_kind = SNMF_SPARSITY_RVEC;    // :153-154
        sparsity = mxGetDoubles(sp);
    } else if (mxGetM(sp) == r && mxGetN(sp) == T) {
        p.sparsity_kind = SNMF_SPARSITY_FULL;    // :155
        sparsity = mxGetDoubles(sp);
    } else {
        mexErrMsgIdAndTxt("snmf:dim", "sparsity must be a scalar, r x 1 or r x n");
    }

    // MATLAB value semantics: inputs are never modified, outputs are fresh arrays.  The out-of-place entry reads init_w / init_h
    // where they lie, so nothing is duplicated on the host first (the multi-device entry is in/out: it gets copies).
    // opts.devices = [0 1 ... 7]: the same call over several GPUs (run_basis_DNMF.m:40,47,53 / run_basis_train.m:88 at scale)
    std::vector<int32_t> devices;
    if (const mxArray* dv = mxGetField(opts, 0, "devices")) {
        if (!mxIsEmpty(dv)) {
            if (!mxIsDouble(dv)) mexErrMsgIdAndTxt("snmf:type", "opts.devices must be a double vector of device ordinals");
            const double* d = mxGetDoubles(dv);
            for (size_t i = 0; i < mxGetNumberOfElements(dv); ++i) devices.push_back((int32_t)d[i]);
        }
    }
    mxArray* hout = devices.empty() ? mxCreateDoubleMatrix((mwSize)r, (mwSize)T, mxREAL) : mxDuplicateArray(h0);
    plhs[0] = devices.empty() ? mxCreateDoubleMatrix((mwSize)F, (mwSize)r, mxREAL) : mxDuplicateArray(w0);
    mxArray* divv = mxCreateDoubleMatrix(1, p.max_iter > 0 ? p.max_iter : 1, mxREAL);
    mxArray* costv = mxCreateDoubleMatrix(1, p.max_iter > 0 ? p.max_iter : 1, mxREAL);
    int32_t n_iter = 0;
    const int st = devices.empty()
        ? snmf_sparse_nmf_oop_f64(g_ctx, &p, mxGetDoubles(v), (int64_t)F, mxGetDoubles(w0), mxGetDoubles(h0), sparsity,
                                  mxGetDoubles(plhs[0]), mxGetDoubles(hout), mxGetDoubles(divv), mxGetDoubles(costv), &n_iter)
        : snmf_sparse_nmf_multi_f64(devices.data(), (int32_t)devices.size(), &p, mxGetDoubles(v), (int64_t)F,
                                    mxGetDoubles(plhs[0]), mxGetDoubles(hout), sparsity, mxGetDoubles(divv),
                                    mxGetDoubles(costv), &n_iter);
    if (st != SNMF_OK) {
        mxDestroyArray(hout);
        mxDestroyArray(divv);
        mxDestroyArray(costv);
        mexErrMsgIdAndTxt("snmf:solve", "%s", snmf_last_error());
    }
    if (nlhs > 1) plhs[1] = hout; else mxDestroyArray(hout);
    if (nlhs > 2) plhs[2] = divv; else mxDestroyArray(divv);
    if (nlhs > 3) plhs[3] = costv; else mxDestroyArray(costv);
    if (nlhs > 4) plhs[4] = mxCreateDoubleScalar((double)n_iter);
}
