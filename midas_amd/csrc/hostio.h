#pragma once
#include "../../include/midas_snps.h"
