// Stand-in: block-copy dispatch is not used by the kernels compiled here.
#pragma once
