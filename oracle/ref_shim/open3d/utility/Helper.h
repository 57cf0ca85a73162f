// Stand-in: nothing from open3d/utility/Helper.h is used by the hot-path kernels.
#pragma once
