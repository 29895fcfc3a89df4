#pragma once
#include "../../minibwa.h"
