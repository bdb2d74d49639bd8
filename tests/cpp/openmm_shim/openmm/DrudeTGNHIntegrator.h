// NOT OpenMM (README.md in tests/cpp/openmm_shim): includes the one header of declarations-only stand-ins
#include "../shim_common.h"
