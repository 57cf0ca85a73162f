// Stand-in: nothing from open3d/utility/Preprocessor.h is used by the hot-path kernels.
#pragma once
