{-# LANGUAGE RecordWildCards, ScopedTypeVariables, TypeApplications #-}
-- | Drop-in for Bulletproofs.ArithmeticCircuit (Bulletproofs/ArithmeticCircuit.hs:1-10): `generateProof` and
-- `verifyProof` with the reference's types (ArithmeticCircuit/Prover.hs:16-21, Verifier.hs:16-21).  The circuit goes
-- down UNPADDED as dense row-major q x n matrices; padCircuit / padAssignment (Internal.hs:76-97) happen inside the
-- engine.  Draw order of the reference: aiBlinding, aoBlinding, sBlinding (Prover.hs:25-27), chooseBlindingVectors n
-- on the PADDED n (:32), five tBlindings (:57).  NOT COMPILED here (no GHC in the build image).
module Bulletproofs.Gpu.ArithmeticCircuit
  ( generateProof
  , verifyProof
  , module Bulletproofs.ArithmeticCircuit.Internal
  ) where

import Protolude
import qualified Data.ByteString as BS
import qualified Data.ByteString.Internal as BI
import Control.Monad.Random (MonadRandom, getRandomR)
import Foreign (withForeignPtr)
import System.IO.Unsafe (unsafePerformIO)

import Data.Curve.Weierstrass.SECP256K1 (PA, Fr, _r)

import Bulletproofs.ArithmeticCircuit.Internal
import Bulletproofs.InnerProductProof.Internal (InnerProductProof(..))
import Bulletproofs.Utils (chooseBlindingVectors, log2Ceil)
import Bulletproofs.Gpu.FFI

-- dense row-major matrix of canonical scalars
encMatrix :: [[Fr]] -> BS.ByteString
encMatrix = BS.concat . fmap encFrs

-- (unpadded gate count, q, m) of a circuit and its assignment
dims :: ArithCircuit Fr -> Int -> (Int, Int, Int)
dims ArithCircuit{..} nAssignment =
  let GateWeights{..} = weights
      q = length wL
      n = maybe nAssignment length (head wL)
      m = maybe 0 length (head commitmentWeights)
  in (n, q, m)

padded :: Int -> Int
padded 0 = 0
padded n = 2 ^ log2Ceil n

generateProof
  :: forall m . (MonadRandom m)
  => ArithCircuit Fr
  -> ArithWitness Fr PA
  -> m (ArithCircuitProof Fr PA)
generateProof circuit@ArithCircuit{..} ArithWitness{..} = do
  let GateWeights{..} = weights
      Assignment{..} = assignment
      (n, q, mV) = dims circuit (length aL)
      n2 = padded n
      k = if n2 <= 1 then 0 else log2Ceil n2
      genBlinding = (fromInteger :: Integer -> Fr) <$> getRandomR (1, fromIntegral _r - 1)
  aiBlinding <- genBlinding
  aoBlinding <- genBlinding
  sBlinding <- genBlinding
  (sL :: [Fr], sR :: [Fr]) <- chooseBlindingVectors (fromIntegral n2)
  tRandoms <- replicateM 5 ((fromInteger @Fr) <$> getRandomR (1, fromIntegral _r - 1))
  pure $ unsafePerformIO $ do
    let npts = 10 + 2 * k
    sc  <- BI.mallocByteString (5 * 32)
    pts <- BI.mallocByteString (npts * 64)
    withForeignPtr sc $ \psc -> withForeignPtr pts $ \ppt ->
      withBS (encMatrix wL) $ \pwl -> withBS (encMatrix wR) $ \pwr -> withBS (encMatrix wO) $ \pwo ->
      withBS (encMatrix commitmentWeights) $ \pwv ->
      withBS (encFrs aL) $ \pal -> withBS (encFrs aR) $ \par -> withBS (encFrs aO) $ \pao ->
      withBS (encFrs commitBlinders) $ \pcb ->
      withBS (encFrs [aiBlinding, aoBlinding, sBlinding]) $ \pb3 ->
      withBS (encFrs sL) $ \psl -> withBS (encFrs sR) $ \psr -> withBS (encFrs tRandoms) $ \pt5 ->
        withCtx (\cx -> c_ac_prove cx (fromIntegral n) (fromIntegral q) (fromIntegral mV) pwl pwr pwo pwv pal par pao pcb pb3 psl psr pt5
                   psc ppt >>= check cx "bp_ac_prove")
    let scalars = BI.fromForeignPtr sc 0 (5 * 32)
        points  = BI.fromForeignPtr pts 0 (npts * 64)
        sAt i = decFr (BS.take 32 (BS.drop (32 * i) scalars))
        pAt i = decPt (BS.take 64 (BS.drop (64 * i) points))
    pure ArithCircuitProof
      { tBlinding = sAt 0, mu = sAt 1, t = sAt 2
      , aiCommit = pAt 0, aoCommit = pAt 1, sCommit = pAt 2, tCommits = pAt <$> [3 .. 9]
      , productProof = InnerProductProof
          { lCommits = pAt <$> [10 .. 9 + k], rCommits = pAt <$> [10 + k .. 9 + 2 * k], l = sAt 3, r = sAt 4 } }

verifyProof
  :: [PA]
  -> ArithCircuitProof Fr PA
  -> ArithCircuit Fr
  -> Bool
verifyProof vCommits ArithCircuitProof{..} circuit@ArithCircuit{..} = unsafePerformIO $ do
  let GateWeights{..} = weights
      (n, q, mV) = dims circuit 0
      InnerProductProof{..} = productProof
      scalars = encFrs [tBlinding, mu, t, l, r]
      points  = encPts ([aiCommit, aoCommit, sCommit] <> tCommits <> lCommits <> rCommits)
  out <- BI.create 1 $ \po ->
    withBS (encMatrix wL) $ \pwl -> withBS (encMatrix wR) $ \pwr -> withBS (encMatrix wO) $ \pwo ->
    withBS (encMatrix commitmentWeights) $ \pwv -> withBS (encFrs cs) $ \pcs ->
    withBS (encPts (take mV vCommits)) $ \pvc -> withBS scalars $ \psc -> withBS points $ \ppt ->
      withCtx (\cx -> c_ac_verify cx (fromIntegral n) (fromIntegral q) (fromIntegral mV) pwl pwr pwo pwv pcs pvc psc ppt po
        >>= check cx "bp_ac_verify")
  pure (BS.head out /= 0)
