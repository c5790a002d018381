// complex64 instantiations of the mapped row FFT kernel (N = 8 .. 32768)
#include "fft_rows_impl.h"
namespace swf {
int launch_fft_rows(int logn, const RowsArgs<float>& a, const OffTab& tab, hipStream_t s) {
    return Dispatch<float, kMinLogN, kMaxLogNFloat>::launch(logn, a, tab, s);
}
int init_fft_rows_f32() { return Dispatch<float, kMinLogN, kMaxLogNFloat>::init(); }
}  // namespace swf
