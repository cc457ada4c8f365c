// explicit instantiation of the fused EKF-SLAM step kernel: n <= 43, 2 wavefronts per filter, fp32 storage
#include "ekf_kernel_impl.h"
namespace slam {
template hipError_t launch_variant<43, 2, 4, 4, float>(const EkfStepParams&, hipStream_t);
}
