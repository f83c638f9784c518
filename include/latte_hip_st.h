/* ST-voxel aggregation entry points (HR_Block / CR_Block); included by latte_hip.h */
#ifndef LATTE_HIP_ST_H
#define LATTE_HIP_ST_H
#endif
