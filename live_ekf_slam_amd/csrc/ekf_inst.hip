// ekf_inst.hip — ONE explicit instantiation of the fused EKF-SLAM step kernel (ekf_kernel_impl.h), selected by -D flags so
// that the variants compile in parallel (build.py): V_NMAX (state capacity: sizes the LDS arrays), V_W (wavefronts per
// filter), V_KG (detections per group), V_UNR (rows per strip of the bulk stream), V_F32 (fp32 storage of x and P),
// V_PIPE (software-pipelined stream), V_KP (optional: landmark slot pairs).  The unit registers its launcher with the dispatcher in ekf_kernel.hip.
#include "ekf_kernel_impl.h"

#ifndef V_KP
#define V_KP 0   // landmark slot pairs of the thin rows / cols; 0 = derived from KG (EkfGeom)
#endif
#if !defined(V_NMAX) || !defined(V_W) || !defined(V_KG) || !defined(V_UNR) || !defined(V_F32) || !defined(V_PIPE)
#error "compile with -DV_NMAX -DV_W -DV_KG -DV_UNR -DV_F32 -DV_PIPE (live_ekf_slam_amd/build.py)"
#endif

namespace slam {
namespace {
#if V_F32
typedef float StorageT;
#else
typedef double StorageT;
#endif
EkfVariant g_variant = {V_NMAX, V_KP * 10000 + V_PIPE * 1000 + V_W * 100 + V_KG * 10 + V_UNR, V_F32,
                        &launch_variant<V_NMAX, V_W, V_KG, V_UNR, StorageT, V_PIPE, V_KP>,
                        &variant_info<V_NMAX, V_W, V_KG, V_UNR, StorageT, V_PIPE, V_KP>, nullptr};
struct Registrar {
    Registrar() { register_ekf_variant(&g_variant); }
} g_registrar;
}  // namespace
}  // namespace slam
