// complex128 instantiations of the mapped row FFT kernel (N = 8 .. 8192)
#include "fft_rows_impl.h"
namespace swf {
int launch_fft_rows(int logn, const RowsArgs<double>& a, const OffTab& tab, hipStream_t s) {
    return Dispatch<double, kMinLogN, kMaxLogNDouble>::launch(logn, a, tab, s);
}
int init_fft_rows_f64() { return Dispatch<double, kMinLogN, kMaxLogNDouble>::init(); }
}  // namespace swf
