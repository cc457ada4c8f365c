// explicit instantiation of the fused EKF-SLAM step kernel: n <= 103, 4 wavefronts per filter, fp32 storage
#include "ekf_kernel_impl.h"
namespace slam {
template hipError_t launch_variant<103, 4, 4, 4, float>(const EkfStepParams&, hipStream_t);
}
