// snmf_mdi_mex.cpp -- MATLAB MEX shim for the missing-data-imputation variants of the solver
//     [v_MDI, h, objective] = snmf_mdi(v, Dm, p)       src/snmf_mdi.m:1      (binary mask, 1 = observed)
//     [v_MDI, h, objective] = snmf_mdi_Sm(v, Sm, p)    src/snmf_mdi_Sm.m:1   (soft mask in [0,1])
// over the plan API of libsnmf_hip.so (include/snmf.h: snmf_plan_set_mask_f64, snmf_plan_get_v_mdi_f64).
//
//     [v_mdi, w, h, div, cost, n_iter] = snmf_mdi_mex(v, mask, w0, h0, sparsity, opts)
//   v, mask  F x T double;  w0 F x r (init_w, :116-131);  h0 r x T (init_h, :133-140);  sparsity: scalar (p.sparsity_mdi)
//   opts     struct: beta, max_iter, conv_eps (p.conv_eps_mdi), cost_check, w_update_ind, h_update_ind (r x 1), device
// The MATLAB wrapper that shadows src/snmf_mdi.m applies the reference's defaults and draws the random factors with
// MATLAB's own generator, exactly like integration/sparse_nmf.m does for the plain solver.
//
// Written against the documented MEX C API; only SYNTAX-CHECKED here (integration/mex_stub/mex.h), MATLAB being absent.
//     mex -R2018a -I<repo>/include integration/snmf_mdi_mex.cpp -L<repo>/se_snmf_nat_amd -lsnmf_hip
#include <cstdint>
#include <cstring>
#include <vector>

#include "mex.h"
#include "snmf.h"

static snmf_ctx* g_ctx = nullptr;
static int g_device = -1;

static void at_exit() {
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
static void check2d(const mxArray* a, const char* what) {
    if (!mxIsDouble(a) || mxIsComplex(a) || mxGetNumberOfDimensions(a) != 2)
        mexErrMsgIdAndTxt("snmf:type", "%s must be a real double matrix", what);
}
static void fill_mask(const mxArray* opts, const char* name, size_t r, std::vector<uint8_t>& out) {
    out.assign(r, 1);
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
    if (nrhs != 6) mexErrMsgIdAndTxt("snmf:nargin", "usage: [v_mdi,w,h,div,cost,n_iter] = snmf_mdi_mex(v,mask,w0,h0,sparsity,opts)");
    if (nlhs > 6) mexErrMsgIdAndTxt("snmf:nargout", "too many outputs");
    const mxArray *v = prhs[0], *mk = prhs[1], *w0 = prhs[2], *h0 = prhs[3], *sp = prhs[4], *opts = prhs[5];
    check2d(v, "v");
    check2d(mk, "mask");
    check2d(w0, "init_w");
    check2d(h0, "init_h");
    if (!mxIsStruct(opts)) mexErrMsgIdAndTxt("snmf:type", "opts must be a struct");
    const size_t F = mxGetM(v), T = mxGetN(v), r = mxGetN(w0);
    if (mxGetM(mk) != F || mxGetN(mk) != T) mexErrMsgIdAndTxt("snmf:dim", "mask must have the size of v");
    if (mxGetM(w0) != F || mxGetM(h0) != r || mxGetN(h0) != T) mexErrMsgIdAndTxt("snmf:dim", "init_w must be F x r and init_h r x T");
    const int device = (int)opt_scalar(opts, "device", 0);
    if (!g_ctx || g_device != device) {
        at_exit();
        if (snmf_ctx_create(&g_ctx, device) != SNMF_OK) mexErrMsgIdAndTxt("snmf:device", "%s", snmf_last_error());
        g_device = device;
        mexLock();
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
    p.floor_v = 1;
    p.sparsity_kind = SNMF_SPARSITY_SCALAR;
    p.sparsity_scalar = mxIsEmpty(sp) ? 0.0 : mxGetScalar(sp);
    std::vector<uint8_t> wi, hi;
    fill_mask(opts, "w_update_ind", r, wi);
    fill_mask(opts, "h_update_ind", r, hi);
    p.w_update_ind = wi.data();
    p.h_update_ind = hi.data();

    snmf_plan* pl = nullptr;
    if (snmf_plan_create(g_ctx, &p, &pl) != SNMF_OK) mexErrMsgIdAndTxt("snmf:plan", "%s", snmf_last_error());
    int st = SNMF_OK;
    // lazy: a call is only MADE while no earlier one has failed, so the first failure's status AND message survive
#define STEP(expr) do { if (st == SNMF_OK) st = (expr); } while (0)
    STEP(snmf_plan_set_mask_f64(pl, mxGetDoubles(mk), (int64_t)F, 0));  // before set_v: turns the plan into an MDI solve
    STEP(snmf_plan_set_v_f64(pl, mxGetDoubles(v), (int64_t)F, 0));
    STEP(snmf_plan_set_w_f64(pl, mxGetDoubles(w0), (int64_t)F, 0));
    STEP(snmf_plan_set_h_f64(pl, mxGetDoubles(h0), (int64_t)r, 0));
    STEP(snmf_plan_init(pl));
    int32_t n_iter = 0;
    STEP(snmf_plan_run(pl, p.max_iter, &n_iter));
    mxArray* vm = mxCreateDoubleMatrix(F, T, mxREAL);
    mxArray* wout = mxCreateDoubleMatrix(F, r, mxREAL);
    mxArray* hout = mxCreateDoubleMatrix(r, T, mxREAL);
    mxArray* divv = mxCreateDoubleMatrix(1, p.max_iter > 0 ? p.max_iter : 1, mxREAL);
    mxArray* costv = mxCreateDoubleMatrix(1, p.max_iter > 0 ? p.max_iter : 1, mxREAL);
    STEP(snmf_plan_get_v_mdi_f64(pl, mxGetDoubles(vm), (int64_t)F, 0));
    STEP(snmf_plan_get_w_f64(pl, mxGetDoubles(wout), (int64_t)F, 0));
    STEP(snmf_plan_get_h_f64(pl, mxGetDoubles(hout), (int64_t)r, 0));
    STEP(snmf_plan_get_objective(pl, mxGetDoubles(divv), mxGetDoubles(costv), &n_iter));
    snmf_plan_destroy(pl);
    mxArray* outs[5] = {vm, wout, hout, divv, costv};
    if (st != SNMF_OK) {
        for (mxArray* a : outs) mxDestroyArray(a);
        mexErrMsgIdAndTxt("snmf:solve", "%s", snmf_last_error());
    }
    plhs[0] = vm;
    for (int i = 1; i < 5; ++i) {
        if (nlhs > i) plhs[i] = outs[i];
        else mxDestroyArray(outs[i]);
    }
    if (nlhs > 5) plhs[5] = mxCreateDoubleScalar((double)n_iter);
}
