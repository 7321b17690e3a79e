{-# LANGUAGE NamedFieldPuns #-}
-- | Drop-in for Bulletproofs.InnerProductProof (Bulletproofs/InnerProductProof.hs:1-7): `generateProof` and
-- `verifyProof` with the reference's types, on an arbitrary caller-supplied base.  NOT COMPILED here (no GHC).
module Bulletproofs.Gpu.InnerProductProof
  ( generateProof
  , verifyProof
  , module Bulletproofs.InnerProductProof.Internal
  ) where

import Protolude
import qualified Data.ByteString as BS
import qualified Data.ByteString.Internal as BI
import Foreign (withForeignPtr, plusPtr)
import System.IO.Unsafe (unsafePerformIO)

import Data.Curve.Weierstrass.SECP256K1 (PA, Fr)

import Bulletproofs.InnerProductProof.Internal
import Bulletproofs.Gpu.FFI

-- | InnerProductProof/Prover.hs:17-26.  The engine takes vectors of one length n that is 0 or a power of two, which
-- is how every caller in the reference uses it.
generateProof :: InnerProductBase PA -> PA -> InnerProductWitness Fr -> InnerProductProof Fr PA
generateProof InnerProductBase{ bGs, bHs, bH } commitmentLR InnerProductWitness{ ls, rs } = unsafePerformIO $ do
  let n = length ls
      k = if n <= 1 then 0 else length (takeWhile (< n) (iterate (* 2) 1))
  lBuf <- BI.mallocByteString (64 * max k 1)
  rBuf <- BI.mallocByteString (64 * max k 1)
  lrBuf <- BI.mallocByteString 64
  withForeignPtr lBuf $ \pL -> withForeignPtr rBuf $ \pR -> withForeignPtr lrBuf $ \pl ->
    withBS (encPts (take n bGs)) $ \pg -> withBS (encPts (take n bHs)) $ \ph ->
    withBS (encPt bH) $ \pu -> withBS (encPt commitmentLR) $ \pp ->
    withBS (encFrs ls) $ \pls -> withBS (encFrs (take n rs)) $ \prs ->
      withCtx (\cx -> c_ipp_prove cx (fromIntegral n) pg ph pu pp pls prs pL pR pl (pl `plusPtr` 32) >>= check cx "bp_ipp_prove")
  let lr = BI.fromForeignPtr lrBuf 0 64
  pure InnerProductProof
    { lCommits = decPts k (BI.fromForeignPtr lBuf 0 (64 * k))
    , rCommits = decPts k (BI.fromForeignPtr rBuf 0 (64 * k))
    , l = decFr (BS.take 32 lr)
    , r = decFr (BS.drop 32 lr)
    }

-- | InnerProductProof/Verifier.hs:18-25: `verifyProof n base commitmentLR proof`.
verifyProof :: Integer -> InnerProductBase PA -> PA -> InnerProductProof Fr PA -> Bool
verifyProof n InnerProductBase{ bGs, bHs, bH } commitmentLR InnerProductProof{ lCommits, rCommits, l, r } =
  unsafePerformIO $ do
    let k = min (length lCommits) (length rCommits)   -- zip lCommits rCommits (Verifier.hs:53)
    out <- BI.create 1 $ \po ->
      withBS (encPts bGs) $ \pg -> withBS (encPts bHs) $ \ph ->
      withBS (encPt bH) $ \pu -> withBS (encPt commitmentLR) $ \pp ->
      withBS (encPts (take k lCommits)) $ \pL -> withBS (encPts (take k rCommits)) $ \pR ->
      withBS (encFr l) $ \pl -> withBS (encFr r) $ \pr ->
        withCtx (\cx -> c_ipp_verify cx (fromIntegral n) pg (fromIntegral (length bGs)) ph (fromIntegral (length bHs)) pu pp
                     pL pR (fromIntegral k) pl pr po >>= check cx "bp_ipp_verify")
    pure (BS.head out /= 0)
