#include "speedy_oracle.h"
