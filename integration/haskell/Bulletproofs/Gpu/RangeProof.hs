-- | Drop-in for Bulletproofs.RangeProof (Bulletproofs/RangeProof.hs:1-8): the m = 1 wrappers of
-- RangeProof/Prover.hs:15-32 and RangeProof/Verifier.hs:17-24 with the reference's types.
-- NOT COMPILED here (no GHC in the build image).
module Bulletproofs.Gpu.RangeProof
  ( generateProof
  , generateProofUnsafe
  , verifyProof
  ) where

import Protolude
import Control.Monad.Random (MonadRandom)

import Data.Curve.Weierstrass.SECP256K1 (PA, Fr)

import Bulletproofs.RangeProof.Internal
import qualified Bulletproofs.Gpu.MultiRangeProof as MRP

-- | RangeProof/Prover.hs:15-22.
generateProof
  :: (MonadRandom m)
  => Integer
  -> (Fr, Fr)
  -> ExceptT (RangeProofError Fr) m (RangeProof Fr PA)
generateProof upperBound (v, vBlinding) =
  MRP.generateProof upperBound [(v, vBlinding)]

-- | RangeProof/Prover.hs:25-32.
generateProofUnsafe
  :: (MonadRandom m)
  => Integer
  -> (Fr, Fr)
  -> m (RangeProof Fr PA)
generateProofUnsafe upperBound (v, vBlinding) =
  MRP.generateProofUnsafe upperBound [(v, vBlinding)]

-- | RangeProof/Verifier.hs:17-24.
verifyProof
  :: Integer
  -> PA
  -> RangeProof Fr PA
  -> Bool
verifyProof upperBound vCommit proof
  = MRP.verifyProof upperBound [vCommit] proof
