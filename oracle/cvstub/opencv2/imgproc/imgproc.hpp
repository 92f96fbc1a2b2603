// oracle/cvstub: see opencv2/core/core.hpp (test infrastructure only)
#include <opencv2/core/core.hpp>
