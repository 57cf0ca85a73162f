// Stand-in: nothing from open3d/utility/Timer.h is used by the hot-path kernels.
#pragma once
