#include "../core/core.hpp"
