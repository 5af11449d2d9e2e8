// gut_render_general.hip — the unsorted compositors (gut_render.hip) instantiated for the reference's generalised Gaussian kernels
// other than the quadratic default: render.particle_kernel_degree = 0 (linear), 1 (Laplacian), 3, 4, 5, 8
// (threedgut.cuh:35, particleResponse<> / particleResponseGrd<> in kernels/cuda/models/gaussianParticles.cuh:211-306).
// Its own translation unit so that the default kernels are compiled exactly as without it (see the note at the top of gut_render.hip).
#define GUT_RENDER_GENERAL_TU 1
#undef GUT_CLOCK_STAMPS   // the diagnostic stamps exist in the default unit only
#undef GUT_K7_EXTRA_LDS
#undef GUT_K6_EXTRA_LDS
#include "gut_render.hip"
