// Stand-in for <tbb/concurrent_unordered_map.h> (header presence only).
#pragma once
#include <unordered_map>
