-- | Drop-in for the curve-arithmetic helpers of Bulletproofs.Utils (Bulletproofs/Utils.hs:71-85), same names and
-- types, evaluated by the MI355X engine.  NOT COMPILED here (no GHC in the build image).
module Bulletproofs.Gpu.Utils
  ( sumExps, addTwoMulP, commit, mulPs
  ) where

import Protolude
import qualified Data.ByteString.Internal as BI
import System.IO.Unsafe (unsafePerformIO)

import Data.Curve.Weierstrass.SECP256K1 (PA, Fr, gen)

import Bulletproofs.Utils (h)
import Bulletproofs.Gpu.FFI

-- | `sumExps` (Utils.hs:76-80): truncates to the shorter list, [] -> mempty.
sumExps :: [Fr] -> [PA] -> PA
sumExps es ps = unsafePerformIO $ do
  let n = min (length es) (length ps)
  out <- BI.create 64 $ \po ->
    withBS (encFrs (take n es)) $ \pe ->
    withBS (encPts (take n ps)) $ \pp ->
      withCtx (\cx -> c_msm cx pe pp (fromIntegral n) po >>= check cx "bp_msm")
  pure (decPt out)

-- | `addTwoMulP` (Utils.hs:72-73): e0 * p0 + e1 * p1.
addTwoMulP :: Fr -> PA -> Fr -> PA -> PA
addTwoMulP e0 p0 e1 p1 = unsafePerformIO $ do
  out <- BI.create 64 $ \po ->
    withBS (encFr e0) $ \a -> withBS (encPt p0) $ \b -> withBS (encFr e1) $ \c -> withBS (encPt p1) $ \d ->
      withCtx (\cx -> c_add_two_mul cx a b c d 1 po >>= check cx "bp_add_two_mul")
  pure (decPt out)

-- | `commit x r = addTwoMulP x gen r h` (Utils.hs:84-85): the reference's type, its own `gen` and `h`.
commit :: Fr -> Fr -> PA
commit x r = addTwoMulP x gen r h

-- | zipWith mul: n independent scalar multiplications in one call (the hs' rescale of
-- MultiRangeProof/Prover.hs:107 for callers that want hs' as points).
mulPs :: [PA] -> [Fr] -> [PA]
mulPs ps es = unsafePerformIO $ do
  let n = min (length es) (length ps)
  out <- BI.create (64 * n) $ \po ->
    withBS (encFrs (take n es)) $ \pe -> withBS (encPts (take n ps)) $ \pp ->
      withCtx (\cx -> c_point_mul cx pe pp (fromIntegral n) po >>= check cx "bp_point_mul")
  pure (decPts n out)
