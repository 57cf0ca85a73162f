// Stand-in: nothing from open3d/utility/Overload.h is used by the hot-path kernels.
#pragma once
