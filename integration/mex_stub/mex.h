/* mex.h -- COMPILE-ONLY stub of the documented MathWorks MEX / matrix C API, limited to the entry points the two
 * shims in this directory use.  MATLAB is not installed in the build container, so the real mex.h / matrix.h are
 * absent; __graft_entry__.build() runs `g++ -fsyntax-only -Iintegration/mex_stub` over the *_mex.cpp shims so that
 * a typo or a wrong argument type in a shim cannot ship unnoticed.  Prototypes follow the public API reference
 * ("C Matrix API", "C MEX API", interleaved-complex / -R2018a names).  Nothing links against this file and it is not
 * a substitute for MATLAB's header: build the shims with `mex -R2018a` as INTEGRATION.md says. */
#ifndef SNMF_MEX_STUB_H
#define SNMF_MEX_STUB_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
typedef struct mxArray_tag mxArray;
typedef size_t mwSize;
typedef size_t mwIndex;
typedef bool mxLogical;
typedef double mxDouble;
typedef int16_t mxInt16;
typedef enum { mxUNKNOWN_CLASS = 0, mxLOGICAL_CLASS = 3, mxCHAR_CLASS = 4, mxDOUBLE_CLASS = 6, mxSINGLE_CLASS = 7,
               mxINT16_CLASS = 10, mxINT32_CLASS = 12 } mxClassID;
typedef enum { mxREAL = 0, mxCOMPLEX = 1 } mxComplexity;

void mexFunction(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]);
void mexErrMsgIdAndTxt(const char* identifier, const char* fmt, ...);
void mexLock(void);
int mexAtExit(void (*exit_fcn)(void));

bool mxIsDouble(const mxArray* pa);
bool mxIsComplex(const mxArray* pa);
bool mxIsStruct(const mxArray* pa);
bool mxIsChar(const mxArray* pa);
bool mxIsLogical(const mxArray* pa);
bool mxIsEmpty(const mxArray* pa);
size_t mxGetM(const mxArray* pa);
size_t mxGetN(const mxArray* pa);
mwSize mxGetNumberOfDimensions(const mxArray* pa);
size_t mxGetNumberOfElements(const mxArray* pa);
double mxGetScalar(const mxArray* pa);
mxDouble* mxGetDoubles(const mxArray* pa);
mxInt16* mxGetInt16s(const mxArray* pa);
mxLogical* mxGetLogicals(const mxArray* pa);
mxArray* mxGetField(const mxArray* pa, mwIndex index, const char* fieldname);
int mxGetString(const mxArray* pa, char* buf, mwSize buflen);
mxArray* mxCreateDoubleMatrix(mwSize m, mwSize n, mxComplexity flag);
mxArray* mxCreateDoubleScalar(double value);
mxArray* mxCreateNumericMatrix(mwSize m, mwSize n, mxClassID classid, mxComplexity flag);
mxArray* mxCreateStructMatrix(mwSize m, mwSize n, int nfields, const char** fieldnames);
void mxSetField(mxArray* pa, mwIndex index, const char* fieldname, mxArray* value);
mxArray* mxDuplicateArray(const mxArray* in);
void mxDestroyArray(mxArray* pa);
#ifdef __cplusplus
}
#endif
#endif
