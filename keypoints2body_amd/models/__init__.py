from .smpl_data import (BodyModelFitResult, BodyModelParams, FLAMEData, MANOData, SMPLData, SMPLHData,
                        SMPLXData)

__all__ = ["BodyModelFitResult", "BodyModelParams", "MANOData", "FLAMEData", "SMPLData", "SMPLHData", "SMPLXData"]
