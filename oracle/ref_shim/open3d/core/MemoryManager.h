// Stand-in: MemoryManager is not used by the kernels compiled here.
#pragma once
