// offpolicy_unit.hip -- one translation unit for the off-policy dense engine, replay sampling and the fused-step executor: k_exec (exec.hip) calls
// the op bodies defined in dense.hip, sac.hip and per.hip, and device code is not linked across translation units in this build.
#include "dense.hip"
#include "sac.hip"
#include "train_dense.hip"
#include "per.hip"
#include "exec.hip"
